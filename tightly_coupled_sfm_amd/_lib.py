"""ctypes binding of libtcsfm_hip.so (include/tcsfm.h).  There is NO fallback: if the HIP library is
missing or fails to load, importing the engine raises."""
from __future__ import annotations

import ctypes as C
import os

# Process-wide HIP runtime settings this library's launch structure likes.  Both variables are read by HIP at the process's FIRST HIP call
# and they affect every HIP user of the process (torch, RCCL), not only this library -- so importing the package does NOT touch the
# environment (round 5; it used to set them when absent): export them yourself, or ask for it with TCSFM_SET_ENV_DEFAULTS=1 (then they are
# set when absent, as before).  Without them the package says once what is being left on the table (README "Runtime environment").
#  * HIP_FORCE_DEV_KERNARG=1 -- a refinement is a chain of short dependent kernels: kernel arguments must sit in device memory
#    (with 0 every launch fetches them over PCIe and a B=1 call takes 90 us instead of 72 us).
#  * GPU_MAX_HW_QUEUES=8 -- lanes (several refinements in flight, include/tcsfm.h) are HIP streams, and this ROCm maps a process's
#    streams onto FOUR hardware queues unless told otherwise: with four lanes -- or two lanes beside a sequence call's copy and pack
#    streams -- two of them share a queue and run one after the other (measured: 4 lanes 17 000-18 600 frame-pairs/s on 4 queues,
#    22 300 / 24 800 on 8; a sequence with 16 windows per call on two lanes 13 800 -> 20 700 windows/s).
ENV_DEFAULTS = {"HIP_FORCE_DEV_KERNARG": "1", "GPU_MAX_HW_QUEUES": "8"}
ENV_APPLIED = {}          # variable -> True (set here, in time) / False (set here, but HIP was already initialised: no effect)


def _hip_initialised() -> bool:
    import sys
    t = sys.modules.get("torch")
    return t is not None and getattr(t, "cuda", None) is not None and t.cuda.is_initialized()


def apply_env_defaults():
    """set the two HIP defaults when absent (what TCSFM_SET_ENV_DEFAULTS=1 does at import): for applications that own their process, e.g. bench.py"""
    _apply_env_defaults(force=True)


def _apply_env_defaults(force=False):
    import warnings
    if not force and os.environ.get("TCSFM_SET_ENV_DEFAULTS", "0") in ("", "0"):
        return
    late = _hip_initialised()
    for k, v in ENV_DEFAULTS.items():
        if k not in os.environ:
            os.environ[k] = v
            ENV_APPLIED[k] = not late
    if ENV_APPLIED.get("HIP_FORCE_DEV_KERNARG") is False:
        # too late for this process: HIP read the variable when the caller first touched the GPU
        warnings.warn("tightly_coupled_sfm_amd: the GPU was initialised before this package was imported and HIP_FORCE_DEV_KERNARG was not set; "
                      "a B=1 refinement is a chain of short kernels and runs ~17 % slower with kernel arguments in host memory "
                      "(90 vs 72 us per call measured) -- export HIP_FORCE_DEV_KERNARG=1 or import this package first", RuntimeWarning)
    elif os.environ.get("HIP_FORCE_DEV_KERNARG", "1") != "1":
        warnings.warn("tightly_coupled_sfm_amd: HIP_FORCE_DEV_KERNARG is set to something other than 1: chains of short kernels (a B=1 "
                      "refinement) run ~17 % slower with kernel arguments in host memory", RuntimeWarning)


_HINTED = set()


def hint_env(lanes: int = 1):
    """Engine creation / Engine.set_lanes: say ONCE per process what an unset variable leaves on the table (nothing is changed)"""
    import warnings
    if os.environ.get("HIP_FORCE_DEV_KERNARG") != "1" and "kernarg" not in _HINTED:
        _HINTED.add("kernarg")
        warnings.warn("tightly_coupled_sfm_amd: HIP_FORCE_DEV_KERNARG=1 is not set: chains of short kernels (a B=1 refinement) run ~17 % slower with "
                      "kernel arguments in host memory (90 vs 72 us per call measured) -- export it before the process first touches the GPU, or set "
                      "TCSFM_SET_ENV_DEFAULTS=1 and import this package first", RuntimeWarning)
    if lanes > 2 and "GPU_MAX_HW_QUEUES" not in os.environ and "queues" not in _HINTED:
        _HINTED.add("queues")
        warnings.warn(f"tightly_coupled_sfm_amd: {lanes} lanes requested and GPU_MAX_HW_QUEUES is not set: HIP maps a process's streams onto four hardware "
                      "queues and lanes that share one run one after the other -- export GPU_MAX_HW_QUEUES=8 before the process first touches the GPU",
                      RuntimeWarning)


def warn_if_queues_late(lanes: int):
    """Engine.set_lanes: more than two lanes only run side by side on more than HIP's default four hardware queues"""
    hint_env(lanes)
    if lanes > 2 and ENV_APPLIED.get("GPU_MAX_HW_QUEUES") is False:
        import warnings
        warnings.warn(f"tightly_coupled_sfm_amd: {lanes} lanes requested, but the GPU was initialised before this package was imported and "
                      "GPU_MAX_HW_QUEUES was not set: HIP maps the streams onto four hardware queues and lanes that share one run one "
                      "after the other -- export GPU_MAX_HW_QUEUES=8 or import this package first", RuntimeWarning)


_apply_env_defaults()

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "libtcsfm_hip.so")


class Opts(C.Structure):
    """mirror of struct tcsfm_opts"""
    _fields_ = [("n_iters", C.c_int32), ("solver", C.c_int32), ("param", C.c_int32), ("refine", C.c_int32),
                ("automask", C.c_int32), ("depth_is_disp", C.c_int32), ("host_ptrs", C.c_int32), ("argmin", C.c_int32),
                ("w_l1", C.c_float), ("w_ssim", C.c_float), ("w_dc", C.c_float), ("irls_eps", C.c_float),
                ("lambda0", C.c_float), ("lambda_up", C.c_float), ("lambda_down", C.c_float), ("lambda_min", C.c_float),
                ("min_depth", C.c_float), ("max_depth", C.c_float), ("prior_scale", C.c_float), ("lambda_depth", C.c_float), ("prior_depth", C.c_float), ("window_rule", C.c_int32), ("dense_joint", C.c_int32), ("prior_init", C.c_float), ("depth_param", C.c_int32), ("w_pose_consist", C.c_float), ("w_smooth", C.c_float), ("free_source_depths", C.c_int32)]


SOLVER_GN, SOLVER_LM = 0, 1
PARAM_SE3, PARAM_EULER = 0, 1
REFINE_POSE, REFINE_POSE_SCALE = 0, 1
WINDOW_PAIR, WINDOW_REFERENCE = 0, 1
DEPTH_FULL, DEPTH_QUARTER = 0, 1
STAT_POSE = 4
NSTAT = 10        # cost, cost_photo, n_mask, lambda, pose[6]  (include/tcsfm.h TCSFM_STAT_*)

_P = C.c_void_p
_SIGNATURES = {
    "tcsfm_lane_probe": (C.c_int, [_P, C.POINTER(C.c_int), C.POINTER(C.c_float), C.POINTER(C.c_float)]),
    "tcsfm_debug_check_guards": (C.c_int, [C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "tcsfm_debug_guard_selftest": (C.c_int, [C.POINTER(C.c_int)]),
    "tcsfm_create": (C.c_int, [C.POINTER(_P), C.c_int, C.c_int, C.c_int, C.c_int]),
    "tcsfm_destroy": (None, [_P]),
    "tcsfm_last_error": (C.c_char_p, [_P]),
    "tcsfm_set_stream": (C.c_int, [_P, _P]),
    "tcsfm_use_own_stream": (C.c_int, [_P]),
    "tcsfm_synchronize": (C.c_int, [_P]),
    "tcsfm_default_opts": (None, [C.POINTER(Opts)]),
    "tcsfm_algorithmic_bytes_per_pixel": (C.c_int, [C.POINTER(Opts)]),
    "tcsfm_disp_to_depth": (C.c_int, [_P, C.POINTER(Opts), C.c_int64, _P, _P, _P]),
    "tcsfm_smooth_loss": (C.c_int, [_P, C.POINTER(Opts), C.c_int, _P, _P, _P]),
    "tcsfm_ssim": (C.c_int, [_P, C.POINTER(Opts), C.c_int, _P, _P, _P]),
    "tcsfm_warp": (C.c_int, [_P, C.POINTER(Opts), C.c_int] + [_P] * 9),
    "tcsfm_warp_posenet_input": (C.c_int, [_P, C.POINTER(Opts), C.c_int] + [_P] * 8),
    "tcsfm_photometric": (C.c_int, [_P, C.POINTER(Opts), C.c_int] + [_P] * 12),
    "tcsfm_loss_surface": (C.c_int, [_P, C.POINTER(Opts)] + [_P] * 5 + [C.c_int, _P, _P]),
    "tcsfm_linearize": (C.c_int, [_P, C.POINTER(Opts), C.c_int] + [_P] * 10),
    "tcsfm_linearize_window": (C.c_int, [_P, C.POINTER(Opts), C.c_int, C.c_int] + [_P] * 10),
    "tcsfm_refine_window": (C.c_int, [_P, C.POINTER(Opts), C.c_int, C.c_int] + [_P] * 10),
    "tcsfm_refine": (C.c_int, [_P, C.POINTER(Opts), C.c_int] + [_P] * 10),
    "tcsfm_refine_dense_window": (C.c_int, [_P, C.POINTER(Opts), C.c_int, C.c_int] + [_P] * 9),
    "tcsfm_linearize_dense_window": (C.c_int, [_P, C.POINTER(Opts), C.c_int, C.c_int] + [_P] * 10),
    "tcsfm_linearize_dense_window_sources": (C.c_int, [_P, C.POINTER(Opts), C.c_int, C.c_int] + [_P] * 11),
    "tcsfm_refine_dense": (C.c_int, [_P, C.POINTER(Opts), C.c_int] + [_P] * 9),
    "tcsfm_scale_recovery": (C.c_int, [_P, C.POINTER(Opts), C.c_int, _P, _P, C.c_float, C.c_int, _P, _P, _P, _P]),
    "tcsfm_posenet_create": (C.c_int, [_P, C.c_int, C.POINTER(_P)]),
    "tcsfm_posenet_destroy": (None, [_P]),
    "tcsfm_posenet_load": (C.c_int, [_P, _P, _P, _P, _P, _P, _P]),
    "tcsfm_posenet_forward": (C.c_int, [_P, C.c_int, _P, _P]),
    "tcsfm_solve_pose_iteratively": (C.c_int, [_P, _P, C.c_int, C.c_int, C.c_int] + [_P] * 7),
    "tcsfm_set_lanes": (C.c_int, [_P, C.c_int]),
    "tcsfm_set_graph_replay": (C.c_int, [_P, C.c_int]),
    "tcsfm_graph_replay_counts": (C.c_int, [_P, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "tcsfm_set_coalesce": (C.c_int, [_P, C.c_int]),
    "tcsfm_set_coalesce_lanes": (C.c_int, [_P, C.c_int]),
    "tcsfm_refine_window_queued": (C.c_int, [_P, C.POINTER(Opts), C.c_int, C.c_int] + [_P] * 7),
    "tcsfm_refine_dense_window_queued": (C.c_int, [_P, C.POINTER(Opts), C.c_int, C.c_int] + [_P] * 8),
    "tcsfm_refine_window_scale_queued": (C.c_int, [_P, C.POINTER(Opts), C.c_int, C.c_int] + [_P] * 9),
    "tcsfm_flush": (C.c_int, [_P]),
    "tcsfm_coalesce_counts": (C.c_int, [_P, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "tcsfm_refine_window_async": (C.c_int, [_P, C.c_int, C.POINTER(Opts), C.c_int, C.c_int] + [_P] * 10),
    "tcsfm_refine_dense_window_async": (C.c_int, [_P, C.c_int, C.POINTER(Opts), C.c_int, C.c_int] + [_P] * 9),
    "tcsfm_refine_sequence": (C.c_int, [_P, C.POINTER(Opts), C.c_int, C.c_int] + [_P] * 6 + [C.c_int, C.c_int, C.c_int]),
    "tcsfm_odometry_sequence": (C.c_int, [_P, _P, C.c_int, C.POINTER(Opts), C.c_int, C.c_int] + [_P] * 6 + [C.c_int, C.c_int, C.c_int]),
    "tcsfm_refine_dense_sequence": (C.c_int, [_P, C.POINTER(Opts), C.c_int, C.c_int] + [_P] * 6 + [C.c_int, C.c_int, C.c_int]),
    "tcsfm_lane_wait": (C.c_int, [_P, C.c_int]),
    "tcsfm_lane_synchronize": (C.c_int, [_P, C.c_int]),
    "tcsfm_lane_event": (C.c_int, [_P, C.c_int, C.POINTER(_P)]),
    "tcsfm_stream_wait_event": (C.c_int, [_P, _P, _P]),
    "tcsfm_profile_begin": (C.c_int, [_P]),
    "tcsfm_profile_end": (C.c_int, [_P, _P, _P]),
    "tcsfm_profile_kernel_time": (C.c_int, [_P, _P, _P]),
    "tcsfm_profile_kernel_busy": (C.c_int, [_P, _P, _P]),
    "tcsfm_debug_stamps": (C.c_int, [_P, _P]),
    "tcsfm_debug_trace": (C.c_int, [_P, _P, C.c_int64, _P, C.c_int64]),
    "tcsfm_pose_to_matrix": (None, [_P, _P]),
    "tcsfm_matrix_to_pose": (None, [_P, _P]),
    "tcsfm_se3_exp": (None, [_P, _P]),
    "tcsfm_se3_log": (None, [_P, _P]),
    "tcsfm_se3_mul": (None, [_P, _P, _P]),
    "tcsfm_se3_inv": (None, [_P, _P]),
}
EXPORTS = tuple(_SIGNATURES)

_lib = None


def load() -> C.CDLL:
    """Load the HIP library (once).  Raises RuntimeError when it is absent: build it with
    ``python -m tightly_coupled_sfm_amd.build`` or ``__graft_entry__.build()``."""
    global _lib
    if _lib is not None:
        return _lib
    # torch bundles its own ROCm runtime: import it first so that this library binds to the SAME libamdhip64 the
    # process's tensors live in (loading ours first leaves two runtimes and hipSetDevice then reports "no device").
    import torch  # noqa: F401
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(f"{LIB_PATH} not found: the HIP extension is not built (no CPU fallback exists)")
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in _SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the symbol is not exported
        fn.restype, fn.argtypes = res, args
    _lib = lib
    return lib


def check_guards():
    """(live allocations checked, allocations with a damaged guard band); (-1, 0) when TCSFM_DEBUG_GUARDS is not set (include/tcsfm.h)"""
    n, bad = C.c_int(0), C.c_int(0)
    rc = load().tcsfm_debug_check_guards(C.byref(n), C.byref(bad))
    if rc != 0:
        raise RuntimeError(f"tcsfm_debug_check_guards failed ({rc})")
    return n.value, bad.value


def default_opts(**kw) -> Opts:
    o = Opts()
    load().tcsfm_default_opts(C.byref(o))
    for k, v in kw.items():
        if not hasattr(o, k):
            raise KeyError(f"tcsfm_opts has no field {k!r}")
        setattr(o, k, v)
    return o
