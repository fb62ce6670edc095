"""CPU-only checks of the product boundary: the C-ABI library loads without a GPU, exports every
symbol include/tcsfm.h declares, its host-side SE(3) utilities agree with the oracle, and error
paths that need no device behave.  (No compute calls: those are the -m gpu tests.)"""
import ctypes as C
import os
import re

import numpy as np
import pytest

from conftest import REPO


@pytest.fixture(scope="module")
def lib():
    from tightly_coupled_sfm_amd import build, _lib
    build.build()
    return _lib.load()


def test_header_symbols_exported(lib):
    hdr = open(os.path.join(REPO, "include", "tcsfm.h")).read()
    declared = set(re.findall(r"\b(tcsfm_[a-z0-9_]+)\s*\(", hdr))
    from tightly_coupled_sfm_amd import _lib
    assert declared == set(_lib.EXPORTS), declared ^ set(_lib.EXPORTS)
    for name in declared:
        assert hasattr(lib, name), name


def test_opts_struct_matches_header(lib):
    from tightly_coupled_sfm_amd import _lib
    o = _lib.default_opts()
    assert (o.n_iters, o.solver, o.param, o.refine, o.automask) == (4, 0, 0, 0, 1)
    assert abs(o.w_l1 - 0.15) < 1e-7 and abs(o.w_ssim - 0.85) < 1e-7 and abs(o.max_depth - 2.67) < 1e-6
    assert C.sizeof(_lib.Opts) == 8 * 4 + 13 * 4 + 7 * 4          # 8 int32, 13 float, then window_rule / dense_joint (int32) / prior_init (float) / depth_param (int32) / w_pose_consist, w_smooth (float) / free_source_depths (int32)
    assert (o.window_rule, o.dense_joint, o.depth_param) == (_lib.WINDOW_PAIR, 1, _lib.DEPTH_FULL) and abs(o.prior_init - 0.1) < 1e-7
    # the header's struct, compiled by the C compiler, has the same size and the same offsets of the last fields
    import subprocess, tempfile, os
    src = '#include <stdio.h>\n#include <stddef.h>\n#include "tcsfm.h"\nint main(void){printf("%zu %zu %zu %zu %zu %zu %zu %zu", sizeof(tcsfm_opts), offsetof(tcsfm_opts, prior_depth), offsetof(tcsfm_opts, window_rule), offsetof(tcsfm_opts, dense_joint), offsetof(tcsfm_opts, depth_param), offsetof(tcsfm_opts, w_pose_consist), offsetof(tcsfm_opts, w_smooth), offsetof(tcsfm_opts, free_source_depths));return 0;}'
    with tempfile.TemporaryDirectory() as d:
        open(os.path.join(d, "s.c"), "w").write(src)
        subprocess.check_call(["gcc", "-std=c99", "-I" + os.path.join(REPO, "include"), os.path.join(d, "s.c"), "-o", os.path.join(d, "s")])
        got = [int(x) for x in subprocess.check_output([os.path.join(d, "s")]).split()]
    assert got == [C.sizeof(_lib.Opts), _lib.Opts.prior_depth.offset, _lib.Opts.window_rule.offset, _lib.Opts.dense_joint.offset, _lib.Opts.depth_param.offset, _lib.Opts.w_pose_consist.offset, _lib.Opts.w_smooth.offset, _lib.Opts.free_source_depths.offset]
    assert lib.tcsfm_algorithmic_bytes_per_pixel(C.byref(o)) == 32


def test_se3_host_utilities_match_oracle(lib, oracle64):
    from tightly_coupled_sfm_amd import engine as E
    rng = np.random.default_rng(0)
    for _ in range(20):
        pose = rng.normal(scale=[0.05, 0.05, 0.05, 0.3, 0.3, 0.3])
        T = E.pose_to_matrix(pose)
        assert np.allclose(T, oracle64.pose_to_T(pose), atol=1e-15)
        assert np.allclose(E.matrix_to_pose(T), pose, atol=1e-13)
        xi = rng.normal(scale=[0.1, 0.1, 0.1, 0.5, 0.5, 0.5])
        X = E.se3_exp(xi)
        assert np.allclose(X, oracle64.se3_exp(xi), atol=1e-15)
        assert np.allclose(E.se3_log(X), xi, atol=1e-12)
        assert np.allclose(X[:, :3] @ X[:, :3].T, np.eye(3), atol=1e-14)
        Xi = E.se3_inv(X)
        assert np.allclose(E.se3_mul(X, Xi), np.eye(4)[:3], atol=1e-14)
    # small-angle series branch
    xi = np.array([0.01, -0.02, 0.03, 1e-7, -2e-7, 3e-7])
    assert np.allclose(E.se3_log(E.se3_exp(xi)), xi, atol=1e-15)
    # liegroups convention: translation first, exp of a pure translation twist is that translation
    assert np.allclose(E.se3_exp([1, 2, 3, 0, 0, 0])[:, 3], [1, 2, 3])


def test_euler_chain_rule(oracle64):
    """A = d(xi_left)/d(pose) used by the additive-Euler parameterisation, by finite differences"""
    from tightly_coupled_sfm_amd import engine as E
    pose = np.array([0.02, -0.01, 0.04, 0.2, -0.1, 0.15])
    A = oracle64.euler_left_jacobian(pose)
    T0 = E.pose_to_matrix(pose)
    h = 1e-6
    for j in range(6):
        dp = np.zeros(6); dp[j] = h
        xi = E.se3_log(E.se3_mul(E.pose_to_matrix(pose + dp), E.se3_inv(T0)))
        assert np.allclose(xi / h, A[:, j], atol=2e-6)


def test_create_without_gpu_fails_cleanly(lib):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    h = C.c_void_p()
    assert lib.tcsfm_create(C.byref(h), 0, 192, 640, 2) < 0
    assert h.value is None and b"tcsfm_create" in lib.tcsfm_last_error(None)
    assert lib.tcsfm_create(C.byref(h), 0, 2, 2, 1) == -1       # bad sizes
    from tightly_coupled_sfm_amd.engine import Engine
    with pytest.raises(RuntimeError):
        Engine(192, 640, 1)


def test_product_never_imports_oracle():
    """the oracle is test infrastructure: nothing under the product package may reference it"""
    pkg = os.path.join(REPO, "tightly_coupled_sfm_amd")
    for root, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h")):
                txt = open(os.path.join(root, f)).read()
                assert "import oracle" not in txt and "from oracle" not in txt and "liboracle" not in txt, f


def _build_c_caller(tmp_path):
    import subprocess
    from tightly_coupled_sfm_amd import _lib
    exe = str(tmp_path / "c_caller")
    libdir = os.path.dirname(_lib.LIB_PATH)
    subprocess.check_call(["gcc", "-std=c99", "-O2", "-Wall", "-Wextra", "-pedantic", "-Werror", "-I" + os.path.join(REPO, "include"),
                           os.path.join(REPO, "examples", "c_caller.c"), "-L" + libdir, "-ltcsfm_hip", "-Wl,-rpath," + libdir, "-lm", "-o", exe])
    return exe


def test_header_is_c99_and_a_plain_c_program_links(lib, tmp_path):
    """the boundary is a C ABI: include/tcsfm.h compiles as strict C99 and examples/c_caller.c (no torch, no C++) links
    against the library; without a GPU it fails with the library's own error message, not a crash"""
    import subprocess
    exe = _build_c_caller(tmp_path)
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present: the run itself is covered by tests/test_gpu_parity.py::test_plain_c_caller")
    r = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert r.returncode == 1 and "tcsfm_create" in r.stderr


def test_liegroups_stand_in(lib, oracle64):
    """the SE3 / SO3 slice of liegroups the reference uses, on the library's own SE(3) routines (parity unpinned: checked for
    self-consistency and against the oracle's independent closed forms)"""
    from tightly_coupled_sfm_amd.liegroups import SE3, SO3
    rng = np.random.default_rng(3)
    for scale in (1e-9, 1e-4, 0.3, 2.5):
        xi = rng.normal(size=6) * np.array([1, 1, 1, scale, scale, scale])
        T = SE3.exp(xi)
        assert np.allclose(T.log(), xi, atol=1e-9) and np.allclose(T.as_matrix()[:3], oracle64.se3_exp(xi).reshape(3, 4), atol=1e-12)
        assert np.allclose(T.dot(T.inv()).as_matrix(), np.eye(4), atol=1e-12)
        assert np.allclose(T.rot.as_matrix() @ T.rot.as_matrix().T, np.eye(3), atol=1e-12)
        U = SE3.exp(rng.normal(size=6) * 0.2)
        assert np.allclose(T.dot(U).as_matrix(), T.as_matrix() @ U.as_matrix(), atol=1e-12)
        p = rng.normal(size=(5, 3))
        assert np.allclose(T.dot(p), p @ T.as_matrix()[:3, :3].T + T.trans, atol=1e-12) and np.allclose(T.dot(p[0]), T.dot(p)[0])
        assert np.allclose(SO3.exp(xi[3:]).as_matrix(), T.rot.as_matrix(), atol=1e-12) and np.allclose(SO3.exp(xi[3:]).log(), xi[3:], atol=1e-9)
    noisy = SE3.exp([0.1, 0.2, 0.3, 0.2, -0.1, 0.05]).as_matrix()
    noisy[:3, :3] += 1e-3 * rng.normal(size=(3, 3))
    R = SE3.from_matrix(noisy, normalize=True).rot.as_matrix()
    assert np.allclose(R @ R.T, np.eye(3), atol=1e-12) and abs(np.linalg.det(R) - 1) < 1e-12
    # the way the reference builds its pose vectors (kitti_loader_stereo.py:135-147): log of the relative transform
    A, B = SE3.exp([0.5, 0, 2.0, 0, 0.1, 0]), SE3.exp([0.6, 0.05, 3.0, 0.01, 0.12, 0])
    rel = A.inv().dot(B)
    assert np.allclose(SE3.exp(rel.log()).as_matrix(), rel.as_matrix(), atol=1e-12)


def test_runtime_environment_untouched_by_default():
    """importing the binding leaves the process environment alone (the variables are process-wide: torch and RCCL see them)"""
    import subprocess, sys
    code = ("import os\nfor k in ('HIP_FORCE_DEV_KERNARG', 'GPU_MAX_HW_QUEUES', 'TCSFM_SET_ENV_DEFAULTS'): os.environ.pop(k, None)\n"
            "from tightly_coupled_sfm_amd import _lib\n"
            "print('HIP_FORCE_DEV_KERNARG' in os.environ, 'GPU_MAX_HW_QUEUES' in os.environ, _lib.ENV_APPLIED)")
    out = subprocess.check_output([sys.executable, "-c", code], cwd=REPO, text=True).split()
    assert out == ["False", "False", "{}"]


def test_runtime_environment_defaults_opt_in():
    """TCSFM_SET_ENV_DEFAULTS=1 (or _lib.apply_env_defaults()) sets the two HIP runtime defaults the launch structure likes -- when the caller
    has not set them; a caller's own setting is kept"""
    import subprocess, sys
    code = ("import os\nfor k in ('HIP_FORCE_DEV_KERNARG', 'GPU_MAX_HW_QUEUES'): os.environ.pop(k, None)\n"
            "os.environ['TCSFM_SET_ENV_DEFAULTS'] = '1'\n"
            "from tightly_coupled_sfm_amd import _lib\nprint(os.environ['HIP_FORCE_DEV_KERNARG'], os.environ['GPU_MAX_HW_QUEUES'])\n"
            "os.environ['GPU_MAX_HW_QUEUES'] = '2'\nimport importlib; importlib.reload(_lib)\nprint(os.environ['GPU_MAX_HW_QUEUES'])\n"
            "del os.environ['TCSFM_SET_ENV_DEFAULTS']; os.environ.pop('HIP_FORCE_DEV_KERNARG')\n_lib.apply_env_defaults(); print(os.environ['HIP_FORCE_DEV_KERNARG'])")
    out = subprocess.check_output([sys.executable, "-c", code], cwd=REPO, text=True).split()
    assert out == ["1", "8", "2", "1"]


def test_engine_set_lanes_updates_attribute():
    """Engine.set_lanes keeps Engine.lanes current (the assignment had slipped behind a return)"""
    import inspect
    from tightly_coupled_sfm_amd import engine
    src = inspect.getsource(engine.Engine.set_lanes)
    assert "self.lanes = int(n)" in src and src.index("tcsfm_set_lanes") < src.index("self.lanes = int(n)")
    assert "self.lanes" not in inspect.getsource(engine.Engine.graph_replay_counts)


def test_shim_maps_the_references_loss_switches_to_opts():
    """DepthOptimizer._opts(): the reference's option keys select the terms of the engine's cost (no GPU needed: the options struct only)"""
    import warnings
    import torch
    from tightly_coupled_sfm_amd import _lib
    from tightly_coupled_sfm_amd.optimizer import DepthOptimizer
    net = torch.nn.Identity()
    base = {"epochs": 20, "diff_img_argmin": True, "automasking": True, "l_depth_consist": True, "l_depth_consist_weight": 0.15, "l_inverse_reconstruction": True,
            "num_source_imgs": 2, "mode": "scaled", "optimize_depth_pred": False, "plotting": False}
    cfg = {"min_depth": 0.06, "max_depth": 2.67, "iterations": 2, "minibatch": 1, "camera_height": 1.65}
    with warnings.catch_warnings():
        warnings.simplefilter("error")          # l_pose_consist is a term of the cost now: no "ignored" warning
        o = DepthOptimizer(dict(base, l_pose_consist=True), cfg, net, net, "09_02")._opts()
    assert abs(o.w_pose_consist - 0.1) < 1e-7 and o.window_rule == _lib.WINDOW_REFERENCE and abs(o.w_dc - 0.15) < 1e-7
    assert DepthOptimizer(base, cfg, net, net, "09_02")._opts().w_pose_consist == 0.0
    with pytest.warns(UserWarning):             # ... except where the engine cannot honour it: the per-pair rule
        o = DepthOptimizer(dict(base, l_pose_consist=True, window_rule="pair"), cfg, net, net, "09_02")._opts()
    assert o.w_pose_consist == 0.0
    with pytest.warns(UserWarning):             # l_smooth acts on the depth map: not a term of the pose-only mode
        DepthOptimizer(dict(base, l_smooth=True), cfg, net, net, "09_02")
    od = DepthOptimizer(dict(base, optimize_depth_pred=True, l_depth_init=True, l_depth_init_weight=0.1), cfg, net, net, "09_02")._opts()
    assert od.window_rule == _lib.WINDOW_REFERENCE and abs(od.prior_init - 0.1) < 1e-7 and od.w_pose_consist == 0.0 and od.w_smooth == 0.0
    with warnings.catch_warnings():
        warnings.simplefilter("error")          # ... and IS one of the pose + depth mode on the reference's loss
        od = DepthOptimizer(dict(base, optimize_depth_pred=True, l_smooth=True, l_smooth_weight=2), cfg, net, net, "09_02")._opts()
    assert abs(od.w_smooth - 2.0) < 1e-7
