"""Host-side engine: torch tensors (device memory, streams) over the C ABI of libtcsfm_hip.so.

PyTorch is plumbing here -- it owns device buffers and the current stream; every computation happens
in the HIP library.  All array arguments are passed as raw device pointers (``tensor.data_ptr()``).

Naming follows the reference: a *pair* is a directed (target, source) frame pair; poses are the
reference's 6-vectors ``[tx,ty,tz,rx,ry,rz]`` (models/stn.py:143-158).
"""
from __future__ import annotations

import ctypes as C
from typing import Optional

import numpy as np
import torch

from . import _lib
from ._lib import Opts, default_opts  # noqa: F401


def _chk(t: torch.Tensor, shape, name: str) -> torch.Tensor:
    if not isinstance(t, torch.Tensor):
        raise TypeError(f"{name} must be a torch.Tensor")
    if not t.is_cuda:
        raise ValueError(f"{name} must live on the GPU (got {t.device})")
    if t.dtype != torch.float32:
        raise TypeError(f"{name} must be float32 (got {t.dtype})")
    if tuple(t.shape) != tuple(shape):
        # same wording as the reference's check_sizes (models/stn.py:24-30)
        raise AssertionError("wrong size for {}, expected {}, got  {}".format(name, "x".join(map(str, shape)), list(t.shape)))
    return t.contiguous()


def _copy_opts(o: Opts) -> Opts:
    c = Opts()
    C.memmove(C.byref(c), C.byref(o), C.sizeof(Opts))
    return c


class Engine:
    """One handle = one GPU + one HIP stream (include/tcsfm.h).  ``max_pairs`` directed pairs of HxW."""

    def __init__(self, H: int, W: int, max_pairs: int, device: Optional[int] = None, lanes: int = 1):
        self.lib = _lib.load()
        if not torch.cuda.is_available():
            raise RuntimeError("tightly_coupled_sfm_amd.Engine needs a ROCm GPU (torch.cuda.is_available() is False)")
        self.device = torch.cuda.current_device() if device is None else int(device)
        self.H, self.W, self.max_pairs = int(H), int(W), int(max_pairs)
        h = C.c_void_p()
        rc = self.lib.tcsfm_create(C.byref(h), self.device, self.H, self.W, self.max_pairs)
        if rc != 0:
            raise RuntimeError(f"tcsfm_create failed ({rc}): {self.lib.tcsfm_last_error(None).decode()}")
        self._h = h
        self.lanes = 1
        self.lanes_serial = False
        self.use_torch_stream()
        _lib.hint_env(lanes)
        if lanes > 1:
            self.set_lanes(lanes)

    # -- lifetime ----------------------------------------------------------------------------
    def close(self):
        if getattr(self, "_h", None):
            self.lib.tcsfm_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _call(self, rc: int):
        if rc != 0:
            raise RuntimeError(f"tcsfm error {rc}: {self.lib.tcsfm_last_error(self._h).decode()}")

    def use_torch_stream(self):
        """Run on torch's current stream of this device, so tensor producers/consumers stay ordered.  Every tensor-level
        method re-binds on entry (a `with torch.cuda.stream(s):` block is honoured); `refine_into` does not."""
        self._follow_torch = True
        self._bound = None
        self._bind()

    def _bind(self):
        if not self._follow_torch:
            return
        s = torch.cuda.current_stream(self.device).cuda_stream
        if s != self._bound:
            self._call(self.lib.tcsfm_set_stream(self._h, C.c_void_p(s)))
            self._bound = s

    def use_own_stream(self):
        """Run on the handle's private non-blocking stream; the caller orders it against torch work (synchronize())."""
        self._follow_torch = False
        self._call(self.lib.tcsfm_use_own_stream(self._h))

    def set_stream(self, hip_stream: int):
        """Run on the given HIP stream (a raw hipStream_t, e.g. ``torch.cuda.Stream().cuda_stream``) from now on; the engine stops
        following torch's current stream.  Captured call graphs of the old stream are dropped (tcsfm_set_stream)."""
        self._follow_torch = False
        self._call(self.lib.tcsfm_set_stream(self._h, C.c_void_p(int(hip_stream))))

    def last_error(self) -> str:
        return self.lib.tcsfm_last_error(self._h).decode()

    def synchronize(self):
        self._call(self.lib.tcsfm_synchronize(self._h))

    def profile_begin(self):
        """bracket every kernel launch with HIP events on the handle's stream (bench.py roofline leg)"""
        self._call(self.lib.tcsfm_profile_begin(self._h))

    def profile_end(self):
        """-> {'linearize'|'solve'|'pack': (summed ms, launches) from HIP event pairs,
               'linearize_kernel': (summed ms, launches) from the in-kernel s_memrealtime bracket of the linearisation launches,
               'linearize_busy': (ms during which at least one of them ran, launches)}"""
        ms = (C.c_double * 3)(); cnt = (C.c_int64 * 3)()
        self._call(self.lib.tcsfm_profile_end(self._h, ms, cnt))
        out = {k: (ms[i], cnt[i]) for i, k in enumerate(("linearize", "solve", "pack"))}
        kms = C.c_double(); kn = C.c_int64()
        self._call(self.lib.tcsfm_profile_kernel_time(self._h, C.byref(kms), C.byref(kn)))
        out["linearize_kernel"] = (kms.value, kn.value)
        bms = C.c_double(); bn = C.c_int64()
        self._call(self.lib.tcsfm_profile_kernel_busy(self._h, C.byref(bms), C.byref(bn)))
        out["linearize_busy"] = (bms.value, bn.value)      # union of the launches' intervals: (ms with >= 1 launch running, launches)
        return out

    def trace_begin(self, n_lin: int, n_pairs: int):
        """parity-test hook (tcsfm_debug_trace): record the discrete decisions of the following refine* calls -- per
        linearisation and pair one uint16 per pixel (mask, warp validity, bilinear cell parity, L1 / depth-consistency sign codes:
        include/tcsfm.h) and the LM accept / keep decision"""
        self._trace = (torch.zeros((n_lin, n_pairs, self.H, self.W), dtype=torch.int16, device=self.dev),
                       torch.ones((n_lin, n_pairs), dtype=torch.int32, device=self.dev))
        b, d = self._trace
        self._call(self.lib.tcsfm_debug_trace(self._h, self._p(b), b.numel(), self._p(d), d.numel()))

    def trace_end(self):
        """-> (bits [n_lin,N,H,W] uint16, decide [n_lin,N] int32) as numpy arrays; switches the trace off"""
        torch.cuda.synchronize(self.device)
        self._call(self.lib.tcsfm_debug_trace(self._h, None, 0, None, 0))
        b, d = self._trace
        self._trace = None
        return b.cpu().numpy().view(np.uint16), d.cpu().numpy()

    @property
    def dev(self) -> torch.device:
        return torch.device("cuda", self.device)

    def _p(self, t: Optional[torch.Tensor]):
        if t is None:
            return None
        if t.device.index != self.device:   # a pointer from another GPU would fault inside the kernel: refuse it here
            raise ValueError(f"tensor lives on {t.device}, this Engine was created for cuda:{self.device}")
        return C.c_void_p(t.data_ptr())

    # -- reference-function drop-ins -----------------------------------------------------------
    def disp_to_depth(self, disp: torch.Tensor, min_depth: float, max_depth: float):
        """utils/learning_helpers.py:77-86 -> (scaled_disp, depth)"""
        self._bind()
        d = disp.contiguous()
        if d.dtype != torch.float32 or not d.is_cuda:
            raise TypeError("disp must be a float32 GPU tensor")
        s, z = torch.empty_like(d), torch.empty_like(d)
        o = default_opts(min_depth=min_depth, max_depth=max_depth)
        self._call(self.lib.tcsfm_disp_to_depth(self._h, C.byref(o), d.numel(), self._p(d), self._p(s), self._p(z)))
        return s, z

    def ssim_loss(self, x: torch.Tensor, y: torch.Tensor) -> torch.Tensor:
        """SSIM_Loss.forward (losses.py:27-41) for [N,C,H,W] tensors"""
        self._bind()
        N, Cc = x.shape[0], x.shape[1]
        x = _chk(x, (N, Cc, self.H, self.W), "x"); y = _chk(y, (N, Cc, self.H, self.W), "y")
        out = torch.empty_like(x)
        o = default_opts()
        self._call(self.lib.tcsfm_ssim(self._h, C.byref(o), N * Cc, self._p(x), self._p(y), self._p(out)))
        return out

    def smooth_loss(self, disp: torch.Tensor, img: torch.Tensor) -> float:
        """get_smooth_loss (losses.py:43-61): disp [N,1,H,W], img [N,3,H,W] -> python float"""
        self._bind()
        N = disp.shape[0]
        disp = _chk(disp, (N, 1, self.H, self.W), "disp"); img = _chk(img, (N, 3, self.H, self.W), "img")
        out = C.c_double()
        self._call(self.lib.tcsfm_smooth_loss(self._h, C.byref(default_opts()), N, self._p(disp), self._p(img), C.byref(out)))
        return out.value

    def inverse_warp2(self, img, depth, ref_depth, pose, intrinsics):
        """models/stn.py:234-273 with the reference's argument order; ``pose`` here is what the reference
        passes, i.e. callers that wrote ``inverse_warp2(src, d_t, d_s, -poses, K)`` keep passing ``-poses``."""
        self._bind()
        N = img.shape[0]
        H, W = self.H, self.W
        img = _chk(img, (N, 3, H, W), "img"); depth = _chk(depth, (N, 1, H, W), "depth")
        ref_depth = _chk(ref_depth, (N, 1, H, W), "ref_depth"); intrinsics = _chk(intrinsics, (N, 3, 3), "intrinsics")
        pose6 = _chk(pose[:, 0:6].contiguous(), (N, 6), "pose")
        neg = (-pose6).contiguous()  # the ABI applies pose_vec2mat(-pose) like the reference call sites
        rec = torch.empty_like(img); valid = torch.empty_like(depth); pd = torch.empty_like(depth); cd = torch.empty_like(depth)
        o = default_opts()
        self._call(self.lib.tcsfm_warp(self._h, C.byref(o), N, self._p(img), self._p(depth), self._p(ref_depth), self._p(neg),
                                       self._p(intrinsics), self._p(rec), self._p(valid), self._p(pd), self._p(cd)))
        return rec, valid, pd, cd

    def posenet_input(self, target_img, source_img, target_depth, source_depth, pose, intrinsics):
        """(tgt * valid | img_rec) [N,6,H,W] for the next PoseNet call of the coupled iteration (train_mono.py:73-77);
        `pose` is the estimate so far in the reference convention (the warp uses -pose like train_mono.py:80)."""
        self._bind()
        N = target_img.shape[0]
        H, W = self.H, self.W
        t = _chk(target_img, (N, 3, H, W), "target_img"); s = _chk(source_img, (N, 3, H, W), "source_img")
        dt = _chk(target_depth, (N, 1, H, W), "target_depth"); ds = _chk(source_depth, (N, 1, H, W), "source_depth")
        p = _chk(pose, (N, 6), "pose"); K = _chk(intrinsics, (N, 3, 3), "intrinsics")
        out = torch.empty((N, 6, H, W), device=t.device, dtype=torch.float32)
        o = default_opts()
        self._call(self.lib.tcsfm_warp_posenet_input(self._h, C.byref(o), N, self._p(t), self._p(s), self._p(dt), self._p(ds), self._p(p),
                                                     self._p(K), self._p(out), None))
        return out

    def compute_photometric_error(self, target_img, source_img, target_depth, source_depth, pose, intrinsics,
                                  opts: Optional[Opts] = None):
        """optimization_experiments/helpers.py:8-23 -> dict with the reference's keys (+ the raw maps)."""
        self._bind()
        N = target_img.shape[0]
        H, W = self.H, self.W
        t = _chk(target_img, (N, 3, H, W), "target_img"); s = _chk(source_img, (N, 3, H, W), "source_img")
        dt = _chk(target_depth, (N, 1, H, W), "target_depth"); ds = _chk(source_depth, (N, 1, H, W), "source_depth")
        p = _chk(pose, (N, 6), "pose"); K = _chk(intrinsics, (N, 3, 3), "intrinsics")
        o = opts or default_opts()
        diff, valid, weight, ae, am = (torch.empty_like(dt) for _ in range(5))
        rec = torch.empty_like(t)
        self._call(self.lib.tcsfm_photometric(self._h, C.byref(o), N, self._p(t), self._p(s), self._p(dt), self._p(ds), self._p(p),
                                              self._p(K), self._p(diff), self._p(valid), self._p(weight), self._p(ae), self._p(am),
                                              self._p(rec)))
        return {"diff_img": diff, "img_rec": rec, "valid_mask": am * valid, "weight_mask": weight, "poses": pose,
                "warp_valid": valid, "auto_mask_error": ae, "auto_mask": am}

    def loss_surface(self, target_img, source_img, target_depth, source_depth, intrinsics, poses, opts: Optional[Opts] = None):
        """costs of ONE pair under P candidate poses (plot_loss_surface.py:31-33,45-47) -> np.ndarray [P] float64"""
        self._bind()
        H, W = self.H, self.W
        t = _chk(target_img, (1, 3, H, W), "target_img"); s = _chk(source_img, (1, 3, H, W), "source_img")
        dt = _chk(target_depth, (1, 1, H, W), "target_depth"); ds = _chk(source_depth, (1, 1, H, W), "source_depth")
        K = _chk(intrinsics, (1, 3, 3), "intrinsics")
        P = poses.shape[0]
        p = _chk(poses, (P, 6), "poses")
        o = opts or default_opts()
        out = np.zeros(P, dtype=np.float64)
        self._call(self.lib.tcsfm_loss_surface(self._h, C.byref(o), self._p(t), self._p(s), self._p(dt), self._p(ds), self._p(K),
                                               P, self._p(p), out.ctypes.data_as(C.c_void_p)))
        return out

    # -- Gauss-Newton engine -----------------------------------------------------------------------
    def _pairs(self, tgt, src, depth_t, depth_s, K, pose):
        N = tgt.shape[0]
        H, W = self.H, self.W
        return (N, _chk(tgt, (N, 3, H, W), "tgt"), _chk(src, (N, 3, H, W), "src"), _chk(depth_t, (N, 1, H, W), "depth_t"),
                _chk(depth_s, (N, 1, H, W), "depth_s"), _chk(K, (N, 3, 3), "K"), _chk(pose, (N, 6), "pose"))

    def linearize(self, tgt, src, depth_t, depth_s, K, pose, opts: Optional[Opts] = None, log_scale=None):
        """normal equations at ``pose`` -> dict(H [N,np,np], g [N,np], cost, cost_photo, cost_dc, n_mask [N]) (numpy f64)"""
        self._bind()
        o = opts or default_opts()
        N, tgt, src, depth_t, depth_s, K, pose = self._pairs(tgt, src, depth_t, depth_s, K, pose)
        n_p = 7 if o.refine == _lib.REFINE_POSE_SCALE else 6
        ls = None if log_scale is None else _chk(log_scale, (N,), "log_scale")
        Hm = np.zeros((N, n_p, n_p)); g = np.zeros((N, n_p)); st = np.zeros((N, 4))
        self._call(self.lib.tcsfm_linearize(self._h, C.byref(o), N, self._p(tgt), self._p(src), self._p(depth_t), self._p(depth_s),
                                            self._p(pose), self._p(ls), self._p(K), Hm.ctypes.data_as(C.c_void_p),
                                            g.ctypes.data_as(C.c_void_p), st.ctypes.data_as(C.c_void_p)))
        return dict(H=Hm, g=g, cost=st[:, 0], cost_photo=st[:, 1], cost_dc=st[:, 2], n_mask=st[:, 3])

    def refine(self, tgt, src, depth_t, depth_s, K, pose, opts: Optional[Opts] = None, log_scale=None, stats: bool = False):
        """Refine N directed pairs in place-free style: returns (pose [N,6], log_scale [N] or None, stats or None).
        Asynchronous on the handle's stream; outputs are GPU tensors."""
        self._bind()
        o = opts or default_opts()
        N, tgt, src, depth_t, depth_s, K, pose = self._pairs(tgt, src, depth_t, depth_s, K, pose)
        pose_out = torch.empty_like(pose)
        ls_in = ls_out = None
        if o.refine == _lib.REFINE_POSE_SCALE:
            ls_in = torch.zeros(N, device=pose.device, dtype=torch.float32) if log_scale is None else _chk(log_scale, (N,), "log_scale")
            ls_out = torch.empty_like(ls_in)
        st = torch.empty((N, o.n_iters + 1, _lib.NSTAT), device=pose.device, dtype=torch.float32) if stats else None
        self._call(self.lib.tcsfm_refine(self._h, C.byref(o), N, self._p(tgt), self._p(src), self._p(depth_t), self._p(depth_s),
                                         self._p(K), self._p(pose), self._p(ls_in), self._p(pose_out), self._p(ls_out), self._p(st)))
        return pose_out, ls_out, st

    def linearize_window(self, tgt, srcs, depth_t, depth_s, K, pose, opts: Optional[Opts] = None, log_scale=None, argmin: Optional[bool] = None):
        """ONE linearisation of a window's 2*S*B directed pairs (layouts of refine_window) at ``pose`` under opts.argmin /
        opts.window_rule -> dict(H [2SB,np,np], g [2SB,np], cost, cost_photo, cost_dc, n_mask [2SB]) (numpy f64).  With
        window_rule = WINDOW_REFERENCE the costs add up to the reference's compute_optimization_loss (optimizer.py:47-86)."""
        self._bind()
        o = opts or default_opts()
        if argmin is not None:
            o = _copy_opts(o); o.argmin = 1 if argmin else 0
        S, B = int(srcs.shape[0]), int(srcs.shape[1])
        N = 2 * S * B
        tgt = _chk(tgt, (B, 3, self.H, self.W), "tgt"); srcs = _chk(srcs, (S, B, 3, self.H, self.W), "srcs")
        depth_t = _chk(depth_t, (B, 1, self.H, self.W), "depth_t"); depth_s = _chk(depth_s, (S, B, 1, self.H, self.W), "depth_s")
        K = _chk(K, (B, 3, 3), "K"); pose = _chk(pose, (N, 6), "pose")
        n_p = 7 if o.refine == _lib.REFINE_POSE_SCALE else 6
        ls = None if log_scale is None else _chk(log_scale, (N,), "log_scale")
        Hm = np.zeros((N, n_p, n_p)); g = np.zeros((N, n_p)); st = np.zeros((N, 4))
        self._call(self.lib.tcsfm_linearize_window(self._h, C.byref(o), B, S, self._p(tgt), self._p(srcs), self._p(depth_t), self._p(depth_s),
                                                   self._p(K), self._p(pose), self._p(ls), Hm.ctypes.data_as(C.c_void_p),
                                                   g.ctypes.data_as(C.c_void_p), st.ctypes.data_as(C.c_void_p)))
        return dict(H=Hm, g=g, cost=st[:, 0], cost_photo=st[:, 1], cost_dc=st[:, 2], n_mask=st[:, 3])

    def refine_window(self, tgt, srcs, depth_t, depth_s, K, pose, opts: Optional[Opts] = None, log_scale=None, stats: bool = False,
                      argmin: Optional[bool] = None):
        """Window form (the call surface of solve_pose_iteratively, train_mono.py:41-62): tgt [B,3,H,W], srcs [S,B,3,H,W] (or a
        list of S tensors), depth_t [B,1,H,W], depth_s [S,B,1,H,W] (or list), K [B,3,3], pose [2*S*B,6] in the stacked order
        (forward pairs source-major, then inverse pairs).  The 2*S*B directed pairs are formed inside the library; with
        argmin (default: opts.argmin) and S > 1 the forward pairs use the per-pixel min over the sources (optimizer.py:47-69).
        -> (pose [2SB,6], log_scale [2SB] or None, stats or None)"""
        self._bind()
        o = opts or default_opts()
        if argmin is not None:
            o = _copy_opts(o); o.argmin = 1 if argmin else 0
        if isinstance(srcs, (list, tuple)):
            srcs = torch.stack(list(srcs), 0)
        if isinstance(depth_s, (list, tuple)):
            depth_s = torch.stack(list(depth_s), 0)
        S, B = int(srcs.shape[0]), int(srcs.shape[1])
        N = 2 * S * B
        tgt = _chk(tgt, (B, 3, self.H, self.W), "tgt"); srcs = _chk(srcs, (S, B, 3, self.H, self.W), "srcs")
        depth_t = _chk(depth_t, (B, 1, self.H, self.W), "depth_t"); depth_s = _chk(depth_s, (S, B, 1, self.H, self.W), "depth_s")
        K = _chk(K, (B, 3, 3), "K"); pose = _chk(pose, (N, 6), "pose")
        pose_out = torch.empty_like(pose)
        ls_in = ls_out = None
        if o.refine == _lib.REFINE_POSE_SCALE:
            ls_in = torch.zeros(N, device=pose.device, dtype=torch.float32) if log_scale is None else _chk(log_scale, (N,), "log_scale")
            ls_out = torch.empty_like(ls_in)
        st = torch.empty((N, o.n_iters + 1, _lib.NSTAT), device=pose.device, dtype=torch.float32) if stats else None
        self._call(self.lib.tcsfm_refine_window(self._h, C.byref(o), B, S, self._p(tgt), self._p(srcs), self._p(depth_t),
                                                self._p(depth_s), self._p(K), self._p(pose), self._p(ls_in), self._p(pose_out),
                                                self._p(ls_out), self._p(st)))
        return pose_out, ls_out, st

    def refine_dense(self, tgt, src, depth_t, depth_s, K, pose, opts: Optional[Opts] = None, stats: bool = False):
        """Dense mode: refine pose AND per-pixel inverse depth of the target (per-pixel Schur complement).
        -> (pose [N,6], depth [N,1,H,W], stats or None); opts.lambda_depth / opts.prior_depth / min_depth / max_depth apply."""
        self._bind()
        o = opts or default_opts()
        N, tgt, src, depth_t, depth_s, K, pose = self._pairs(tgt, src, depth_t, depth_s, K, pose)
        pose_out, depth_out = torch.empty_like(pose), torch.empty_like(depth_t)
        st = torch.empty((N, o.n_iters + 1, _lib.NSTAT), device=pose.device, dtype=torch.float32) if stats else None
        self._call(self.lib.tcsfm_refine_dense(self._h, C.byref(o), N, self._p(tgt), self._p(src), self._p(depth_t), self._p(depth_s),
                                               self._p(K), self._p(pose), self._p(pose_out), self._p(depth_out), self._p(st)))
        return pose_out, depth_out, st

    def refine_dense_window(self, tgt, srcs, depth_t, depth_s, K, pose, opts: Optional[Opts] = None, stats: bool = False,
                            argmin: Optional[bool] = None):
        """Dense mode in window form (see refine_window for the layout): every directed pair refines its pose and its own copy
        of its target's depth -> (pose [2SB,6], depth [2SB,1,H,W] in the stacked pair order, stats or None)"""
        self._bind()
        o = opts or default_opts()
        if argmin is not None:
            o = _copy_opts(o); o.argmin = 1 if argmin else 0
        if isinstance(srcs, (list, tuple)):
            srcs = torch.stack(list(srcs), 0)
        if isinstance(depth_s, (list, tuple)):
            depth_s = torch.stack(list(depth_s), 0)
        S, B = int(srcs.shape[0]), int(srcs.shape[1])
        N = 2 * S * B
        tgt = _chk(tgt, (B, 3, self.H, self.W), "tgt"); srcs = _chk(srcs, (S, B, 3, self.H, self.W), "srcs")
        depth_t = _chk(depth_t, (B, 1, self.H, self.W), "depth_t"); depth_s = _chk(depth_s, (S, B, 1, self.H, self.W), "depth_s")
        K = _chk(K, (B, 3, 3), "K"); pose = _chk(pose, (N, 6), "pose")
        pose_out = torch.empty_like(pose)
        depth_out = torch.empty((N, 1, self.H, self.W), device=pose.device, dtype=torch.float32)
        st = torch.empty((N, o.n_iters + 1, _lib.NSTAT), device=pose.device, dtype=torch.float32) if stats else None
        self._call(self.lib.tcsfm_refine_dense_window(self._h, C.byref(o), B, S, self._p(tgt), self._p(srcs), self._p(depth_t),
                                                      self._p(depth_s), self._p(K), self._p(pose), self._p(pose_out), self._p(depth_out),
                                                      self._p(st)))
        return pose_out, depth_out, st

    def linearize_dense_window(self, tgt, srcs, depth_t, depth_s, K, pose, opts: Optional[Opts] = None, argmin: Optional[bool] = None, depth0=None,
                               sources: bool = False):
        """tcsfm_linearize_dense_window: ONE linearisation of the dense window mode on the reference's own loss (window_rule REFERENCE;
        optimizer.py:47-90) -> dict(loss, fwd, inv_photo, inv_dc, K_f, K_i, a_f, g_pose [2SB,6] float64 (w.r.t. the left SE(3)
        perturbation of every pair's warp), g_rho [B,1,H,W] GPU tensor = d loss / d inverse depth of every target);
        depth0 [B,1,H,W]: the centre of the l_depth_init prior (default: depth_t).  sources=True (tcsfm_linearize_dense_window_sources):
        also g_rho_src [S,B,1,H,W] = d loss / d inverse depth of every SOURCE map (held fixed by the refinement)"""
        self._bind()
        o = _copy_opts(opts or default_opts())
        if argmin is not None:
            o.argmin = 1 if argmin else 0
        o.window_rule = _lib.WINDOW_REFERENCE
        if isinstance(srcs, (list, tuple)):
            srcs = torch.stack(list(srcs), 0)
        if isinstance(depth_s, (list, tuple)):
            depth_s = torch.stack(list(depth_s), 0)
        S, B = int(srcs.shape[0]), int(srcs.shape[1])
        N = 2 * S * B
        tgt = _chk(tgt, (B, 3, self.H, self.W), "tgt"); srcs = _chk(srcs, (S, B, 3, self.H, self.W), "srcs")
        depth_t = _chk(depth_t, (B, 1, self.H, self.W), "depth_t"); depth_s = _chk(depth_s, (S, B, 1, self.H, self.W), "depth_s")
        K = _chk(K, (B, 3, 3), "K"); pose = _chk(pose, (N, 6), "pose")
        d0 = None if depth0 is None else _chk(depth0, (B, 1, self.H, self.W), "depth0")
        scal = np.zeros(8); gp = np.zeros((N, 6))
        g_rho = torch.empty((B, 1, self.H, self.W), device=pose.device, dtype=torch.float32)
        g_src = torch.empty((S, B, 1, self.H, self.W), device=pose.device, dtype=torch.float32) if sources else None
        if sources:
            self._call(self.lib.tcsfm_linearize_dense_window_sources(self._h, C.byref(o), B, S, self._p(tgt), self._p(srcs), self._p(depth_t), self._p(depth_s),
                                                                     self._p(K), self._p(pose), self._p(d0), scal.ctypes.data_as(C.c_void_p),
                                                                     gp.ctypes.data_as(C.c_void_p), self._p(g_rho), self._p(g_src)))
        else:
            self._call(self.lib.tcsfm_linearize_dense_window(self._h, C.byref(o), B, S, self._p(tgt), self._p(srcs), self._p(depth_t), self._p(depth_s),
                                                             self._p(K), self._p(pose), self._p(d0), scal.ctypes.data_as(C.c_void_p),
                                                             gp.ctypes.data_as(C.c_void_p), self._p(g_rho)))
        torch.cuda.synchronize(self.device)
        out = dict(loss=scal[0], fwd=scal[1], inv_photo=scal[2], inv_dc=scal[3], K_f=scal[4], K_i=scal[5], a_f=scal[6], pose_consist=scal[7], g_pose=gp, g_rho=g_rho)
        if sources:
            out["g_rho_src"] = g_src
        return out

    def scale_recovery(self, depth, intrinsics, real_cam_height: float, pad_to_batch: int = 0, maps: bool = False):
        """ScaleRecovery.forward (dnet_layers.py:306-327): depth [N,1,H,W], K [N,3,3] -> scale [1] (GPU tensor)
        (+ median [1], height [N,1,H,W], mask [N,1,H,W] with maps=True)"""
        self._bind()
        N = depth.shape[0]
        depth = _chk(depth, (N, 1, self.H, self.W), "depth"); K = _chk(intrinsics, (N, 3, 3), "intrinsics")
        scale = torch.empty(1, device=depth.device, dtype=torch.float32); med = torch.empty_like(scale)
        hm = torch.empty_like(depth) if maps else None
        mm = torch.empty_like(depth) if maps else None
        o = default_opts()
        self._call(self.lib.tcsfm_scale_recovery(self._h, C.byref(o), N, self._p(depth), self._p(K), float(real_cam_height),
                                                 int(pad_to_batch), self._p(scale), self._p(med), self._p(hm), self._p(mm)))
        return (scale, med, hm, mm) if maps else scale

    # -- lanes: several refinements in flight (include/tcsfm.h "lanes") -----------------------------------------
    def set_lanes(self, n: int):
        _lib.warn_if_queues_late(int(n))
        self._call(self.lib.tcsfm_set_lanes(self._h, int(n)))
        self.lanes = int(n)
        self.lanes_serial = self.lane_probe()["serial"]

    def lane_probe(self):
        """tcsfm_lane_probe: what tcsfm_set_lanes measured -- {'serial': the streams of this process do not run side by side and lane calls
        fall back to the handle's own stream, 'one_stream_us', 'two_streams_us': the probe's 16 stand-in launches}"""
        s_, a, b = C.c_int(0), C.c_float(0), C.c_float(0)
        self._call(self.lib.tcsfm_lane_probe(self._h, C.byref(s_), C.byref(a), C.byref(b)))
        return {"serial": bool(s_.value), "one_stream_us": round(a.value * 1e3, 1), "two_streams_us": round(b.value * 1e3, 1)}

    def set_graph_replay(self, max_graphs: int = 4):
        """tcsfm_set_graph_replay: repeated device-pointer refine calls (same tensors, same options) are captured once and
        replayed as one HIP graph per call -- one host launch instead of nine; bit-identical results.  0 switches it off."""
        self._call(self.lib.tcsfm_set_graph_replay(self._h, int(max_graphs)))

    def graph_replay_counts(self):
        """(captures, replays) so far, handle + lanes"""
        c, r = C.c_int(0), C.c_int(0)
        self._call(self.lib.tcsfm_graph_replay_counts(self._h, C.byref(c), C.byref(r)))
        return c.value, r.value

    # -- coalesced calls: queued calls of one shape as ONE launch sequence (include/tcsfm.h "coalesced calls") -------------------
    def set_coalesce(self, max_calls: int):
        """tcsfm_set_coalesce: up to `max_calls` queued refine_window_queued calls of one shape run as one launch sequence"""
        self._call(self.lib.tcsfm_set_coalesce(self._h, int(max_calls)))

    def set_coalesce_lanes(self, n_streams: int):
        """tcsfm_set_coalesce_lanes: merged sequences alternate over `n_streams` of the handle's streams (needs that many lanes);
        `flush()` / `synchronize()` order the handle's stream behind them"""
        self._call(self.lib.tcsfm_set_coalesce_lanes(self._h, int(n_streams)))

    def refine_window_queued(self, tgt, srcs, depth_t, depth_s, K, pose, pose_out, opts: Opts):
        """tcsfm_refine_window_queued: note a window call (validated contiguous float32 CUDA tensors in the window layout, see
        refine_window); it runs -- merged with the other waiting calls of its shape -- when the queue is full, at flush() or at
        synchronize().  Per window the result is bit-identical to refine_window on its own."""
        S, B = int(srcs.shape[0]), int(srcs.shape[1])
        self._call(self.lib.tcsfm_refine_window_queued(self._h, C.byref(opts), B, S, self._p(tgt), self._p(srcs), self._p(depth_t), self._p(depth_s),
                                                       self._p(K), self._p(pose), self._p(pose_out)))

    def refine_window_scale_queued(self, tgt, srcs, depth_t, depth_s, K, pose, log_scale, pose_out, log_scale_out, opts: Opts):
        """tcsfm_refine_window_scale_queued: refine_window_queued with the depth-scale unknown (opts.refine = REFINE_POSE_SCALE)"""
        S, B = int(srcs.shape[0]), int(srcs.shape[1])
        self._call(self.lib.tcsfm_refine_window_scale_queued(self._h, C.byref(opts), B, S, self._p(tgt), self._p(srcs), self._p(depth_t), self._p(depth_s),
                                                             self._p(K), self._p(pose), self._p(log_scale), self._p(pose_out), self._p(log_scale_out)))

    def refine_dense_window_queued(self, tgt, srcs, depth_t, depth_s, K, pose, pose_out, depth_out, opts: Opts):
        """tcsfm_refine_dense_window_queued: the dense counterpart of refine_window_queued (window layout of refine_dense_window; depth_out
        [2*S*B,1,H,W]); per-pair Gauss-Newton calls with one source per target are merged, anything else runs at once"""
        S, B = int(srcs.shape[0]), int(srcs.shape[1])
        self._call(self.lib.tcsfm_refine_dense_window_queued(self._h, C.byref(opts), B, S, self._p(tgt), self._p(srcs), self._p(depth_t), self._p(depth_s),
                                                             self._p(K), self._p(pose), self._p(pose_out), self._p(depth_out)))

    def flush(self):
        self._call(self.lib.tcsfm_flush(self._h))

    def coalesce_counts(self):
        """(launch sequences issued, calls they carried)"""
        b, c = C.c_int(0), C.c_int(0)
        self._call(self.lib.tcsfm_coalesce_counts(self._h, C.byref(b), C.byref(c)))
        return b.value, c.value

    def refine_window_async(self, lane: int, tgt, srcs, depth_t, depth_s, K, pose, pose_out, opts: Opts, log_scale=None, log_scale_out=None):
        """tcsfm_refine_window on `lane`, zero-allocation and asynchronous: tensors must be validated / contiguous float32 CUDA
        tensors in the window layout (see refine_window); the lane waits for the work queued on this engine's stream so far.
        Outputs are valid for consumers on this engine's stream after lane_wait(lane), for the host after lane_synchronize(lane)."""
        S, B = int(srcs.shape[0]), int(srcs.shape[1])
        self._call(self.lib.tcsfm_refine_window_async(self._h, int(lane), C.byref(opts), B, S, self._p(tgt), self._p(srcs), self._p(depth_t),
                                                      self._p(depth_s), self._p(K), self._p(pose), self._p(log_scale), self._p(pose_out),
                                                      self._p(log_scale_out), None))

    def refine_dense_window_async(self, lane: int, tgt, srcs, depth_t, depth_s, K, pose, pose_out, depth_out, opts: Opts):
        """tcsfm_refine_dense_window on `lane` (see refine_window_async: validated contiguous float32 CUDA tensors, no allocation;
        pose_out [2SB,6], depth_out [2SB,1,H,W])"""
        S, B = int(srcs.shape[0]), int(srcs.shape[1])
        self._call(self.lib.tcsfm_refine_dense_window_async(self._h, int(lane), C.byref(opts), B, S, self._p(tgt), self._p(srcs), self._p(depth_t),
                                                            self._p(depth_s), self._p(K), self._p(pose), self._p(pose_out), self._p(depth_out), None))

    def refine_sequence(self, frames: torch.Tensor, depths: torch.Tensor, K, init_poses, opts: Optional[Opts] = None, sources: int = 1,
                        ring: int = 0, log_scale: bool = False, windows_per_call: int = 0, target_pos: int = 0):
        """tcsfm_refine_sequence: the whole window loop of a sequence inside the library (frames [T,3,H,W] / depths [T,1,H,W] CPU
        tensors -- pinned for asynchronous copies --, K [3,3], init_poses [T-S, 2S, 6]) -> refined poses [T-S, 2S, 6] (CPU tensor;
        with log_scale=True also the log depth scales [T-S, 2S]); calls of `windows_per_call` windows (0 = default 8, capped by max_pairs / 2S) run on
        the engine's lanes.  Window w = frames w .. w+S; its target is frame w + target_pos (0: the first; -1: the middle one, (S+1)//2, as
        the reference's loaders choose it), its sources the others in order"""
        self._bind()
        o = opts or default_opts()
        cpu = lambda a, shape, name: self._cpu(a, shape, name)
        T, S = int(frames.shape[0]), int(sources)
        frames = cpu(frames, (T, 3, self.H, self.W), "frames"); depths = cpu(depths, (T, 1, self.H, self.W), "depths")
        Kc = cpu(torch.as_tensor(np.asarray(K, dtype=np.float32)), (3, 3), "K")
        p0 = cpu(torch.as_tensor(np.asarray(init_poses, dtype=np.float32)), (T - S, 2 * S, 6), "init_poses")
        out = torch.empty_like(p0)
        ls = torch.zeros((T - S, 2 * S), dtype=torch.float32) if log_scale else None
        hp = lambda t: None if t is None else C.c_void_p(t.data_ptr())
        self._call(self.lib.tcsfm_refine_sequence(self._h, C.byref(o), T, S, hp(frames), hp(depths), hp(Kc), hp(p0), hp(out), hp(ls), int(ring),
                                                  int(windows_per_call), int(target_pos)))
        return (out, ls) if log_scale else out

    def refine_dense_sequence(self, frames: torch.Tensor, depths: torch.Tensor, K, init_poses, opts: Optional[Opts] = None, sources: int = 1,
                              ring: int = 0, windows_per_call: int = 0, target_pos: int = 0, out_depths: Optional[torch.Tensor] = None):
        """tcsfm_refine_dense_sequence: the dense mode (pose + per-pixel inverse depth) over a sequence, arguments as refine_sequence
        -> (poses [T-S, 2S, 6], refined depth maps [T-S, 2S, 1, H, W]) as CPU tensors.
        out_depths: a caller-owned (pinned) CPU tensor of that shape to receive the depth maps; a caller that runs sequence after
        sequence should pass one -- pinning a fresh 70 MB result buffer costs ten times the refinement of a 120-frame sequence."""
        self._bind()
        o = opts or default_opts()
        T, S = int(frames.shape[0]), int(sources)
        frames = self._cpu(frames, (T, 3, self.H, self.W), "frames"); depths = self._cpu(depths, (T, 1, self.H, self.W), "depths")
        Kc = self._cpu(torch.as_tensor(np.asarray(K, dtype=np.float32)), (3, 3), "K")
        p0 = self._cpu(torch.as_tensor(np.asarray(init_poses, dtype=np.float32)), (T - S, 2 * S, 6), "init_poses")
        out = torch.empty_like(p0)
        if out_depths is None:      # pinned: the copies back run beside the kernels
            dout = torch.empty((T - S, 2 * S, 1, self.H, self.W), dtype=torch.float32, pin_memory=True)
        else:
            dout = out_depths
            if dout.is_cuda or dout.dtype != torch.float32 or not dout.is_contiguous() or tuple(dout.shape) != (T - S, 2 * S, 1, self.H, self.W):
                raise TypeError("out_depths must be a contiguous float32 CPU tensor [T-S, 2S, 1, H, W]")
        hp = lambda t: C.c_void_p(t.data_ptr())
        self._call(self.lib.tcsfm_refine_dense_sequence(self._h, C.byref(o), T, S, hp(frames), hp(depths), hp(Kc), hp(p0), hp(out), hp(dout), int(ring),
                                                        int(windows_per_call), int(target_pos)))
        return out, dout

    @staticmethod
    def _cpu(t, shape, name):
        t = torch.as_tensor(t)
        if t.is_cuda or t.dtype != torch.float32:
            raise TypeError(f"{name} must be a float32 CPU tensor (pinned for asynchronous copies)")
        if tuple(t.shape) != tuple(shape):
            raise AssertionError("wrong size for {}, expected {}, got  {}".format(name, "x".join(map(str, shape)), list(t.shape)))
        return t.contiguous()

    def lane_wait(self, lane: int):
        self._call(self.lib.tcsfm_lane_wait(self._h, int(lane)))

    def lane_event(self, lane: int):
        """mark the current end of the lane's work -> opaque event handle for stream_wait_event"""
        ev = C.c_void_p()
        self._call(self.lib.tcsfm_lane_event(self._h, int(lane), C.byref(ev)))
        return ev

    def stream_wait_event(self, stream: "torch.cuda.Stream", event):
        """make a torch stream wait (on the device) for a lane_event mark"""
        self._call(self.lib.tcsfm_stream_wait_event(self._h, C.c_void_p(stream.cuda_stream), event))

    def lane_synchronize(self, lane: int):
        self._call(self.lib.tcsfm_lane_synchronize(self._h, int(lane)))

    def refine_into(self, tgt, src, depth_t, depth_s, K, pose_in, pose_out, opts: Opts, log_scale_in=None, log_scale_out=None,
                    stats_out=None):
        """Zero-allocation variant used by bench.py: tensors must already be validated/contiguous; pose_out may alias pose_in."""
        self._call(self.lib.tcsfm_refine(self._h, C.byref(opts), tgt.shape[0], self._p(tgt), self._p(src), self._p(depth_t),
                                         self._p(depth_s), self._p(K), self._p(pose_in), self._p(log_scale_in), self._p(pose_out),
                                         self._p(log_scale_out), self._p(stats_out)))


# -- SE(3) host utilities (liegroups stand-ins; double precision, no GPU needed) ---------------------
def _vec(a, n):
    a = np.ascontiguousarray(a, dtype=np.float64).reshape(n)
    return a, a.ctypes.data_as(C.c_void_p)


def pose_to_matrix(pose) -> np.ndarray:
    a, pa = _vec(pose, 6); T = np.zeros(12); _lib.load().tcsfm_pose_to_matrix(pa, T.ctypes.data_as(C.c_void_p)); return T.reshape(3, 4)


def matrix_to_pose(T) -> np.ndarray:
    a, pa = _vec(T, 12); p = np.zeros(6); _lib.load().tcsfm_matrix_to_pose(pa, p.ctypes.data_as(C.c_void_p)); return p


def se3_exp(xi) -> np.ndarray:
    a, pa = _vec(xi, 6); T = np.zeros(12); _lib.load().tcsfm_se3_exp(pa, T.ctypes.data_as(C.c_void_p)); return T.reshape(3, 4)


def se3_log(T) -> np.ndarray:
    a, pa = _vec(T, 12); x = np.zeros(6); _lib.load().tcsfm_se3_log(pa, x.ctypes.data_as(C.c_void_p)); return x


def se3_mul(A, B) -> np.ndarray:
    a, pa = _vec(A, 12); b, pb = _vec(B, 12); c = np.zeros(12)
    _lib.load().tcsfm_se3_mul(pa, pb, c.ctypes.data_as(C.c_void_p)); return c.reshape(3, 4)


def se3_inv(A) -> np.ndarray:
    a, pa = _vec(A, 12); b = np.zeros(12); _lib.load().tcsfm_se3_inv(pa, b.ctypes.data_as(C.c_void_p)); return b.reshape(3, 4)
