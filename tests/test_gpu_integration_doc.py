"""INTEGRATION.md shows the ctypes stub a reference maintainer would paste in; this runs that exact text against the
built library, so the documented binding cannot drift from the C ABI."""
import os
import re

import numpy as np
import pytest

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_documented_ctypes_stub_runs_and_matches_engine():
    from tightly_coupled_sfm_amd import _lib, synth
    from tightly_coupled_sfm_amd.engine import Engine, default_opts
    _lib.load()
    text = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    block = re.search(r"```python\n(# tcsfm_binding\.py.*?)```", text, re.S).group(1)
    block = block.replace('C.CDLL("libtcsfm_hip.so")', f'C.CDLL({_lib.LIB_PATH!r})')
    ns = {}
    exec(compile(block, "INTEGRATION.md:tcsfm_binding", "exec"), ns)
    H, W, N = 48, 160, 2
    b = synth.make_batch(N, H, W, seed0=5, both_directions=True)
    t = lambda a: torch.as_tensor(np.ascontiguousarray(a, dtype=np.float32)).cuda()
    tgt, src, dt, ds, K, p0 = (t(b[k]) for k in ("tgt", "src", "depth_t", "depth_s", "K", "pose_init"))
    e = Engine(H, W, N)
    rec, valid, pd, cd = ns["inverse_warp2"](src, dt, ds, -p0, K)          # reference call sites pass -pose (train_mono.py:69)
    rec2, valid2, pd2, cd2 = e.inverse_warp2(src, dt, ds, -p0, K)
    torch.cuda.synchronize()
    assert torch.equal(rec, rec2) and torch.equal(valid, valid2) and torch.equal(pd, pd2) and torch.equal(cd, cd2)
    out = ns["refine_poses"](tgt, src, dt, ds, K, p0, gn_iters=4)
    ref, _, _ = e.refine(tgt, src, dt, ds, K, p0, default_opts(n_iters=4))
    torch.cuda.synchronize()
    assert torch.equal(out, ref)


def test_documented_sequence_stub_runs_and_matches_engine():
    from tightly_coupled_sfm_amd import _lib, synth
    from tightly_coupled_sfm_amd.engine import Engine, default_opts
    _lib.load()
    text = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    block = re.search(r"```python\n(# tcsfm_sequence\.py.*?)```", text, re.S).group(1)
    block = block.replace('C.CDLL("libtcsfm_hip.so")', f'C.CDLL({_lib.LIB_PATH!r})')
    ns = {}
    exec(compile(block, "INTEGRATION.md:tcsfm_sequence", "exec"), ns)
    H, W, T = 48, 160, 14
    seq = synth.make_sequence(T, H, W, seed=4)
    frames, depths = torch.as_tensor(seq["frames"]).pin_memory(), torch.as_tensor(seq["depths"]).pin_memory()
    out = ns["refine_sequence"](frames, depths, seq["K"], seq["init"], sources=1, gn_iters=3, lanes=2)
    ref = Engine(H, W, 2, lanes=2).refine_sequence(frames, depths, seq["K"], seq["init"], default_opts(n_iters=3))
    assert out.shape == (T - 1, 2, 6) and np.array_equal(out, ref.numpy())


def test_documented_dense_reference_stub_runs_and_matches_engine():
    """the documented stub for the dense mode on the reference's own loss (round 4), executed on top of the binding block"""
    from tightly_coupled_sfm_amd import _lib
    from tightly_coupled_sfm_amd.engine import Engine, default_opts
    import test_gpu_dense_reference as T
    _lib.load()
    text = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    base = re.search(r"```python\n(# tcsfm_binding\.py.*?)```", text, re.S).group(1).replace('C.CDLL("libtcsfm_hip.so")', f'C.CDLL({_lib.LIB_PATH!r})')
    dense = re.search(r"```python\n(# the dense mode on the reference's own loss.*?)```", text, re.S).group(1)
    ns = {}
    exec(compile(base, "INTEGRATION.md:tcsfm_binding", "exec"), ns)
    exec(compile(dense, "INTEGRATION.md:dense_reference", "exec"), ns)
    H, W, S = 48, 160, 2
    w = T._window(1, S, H, W, seed=12)
    t = {k: T._dev(v) for k, v in w.items()}
    options = {'diff_img_argmin': True, 'automasking': True, 'l_depth_consist': True, 'l_depth_consist_weight': 0.15, 'l_depth_init': True, 'l_depth_init_weight': 0.1}
    config = {'min_depth': 0.06, 'max_depth': 2.67}
    depths = [t["depth_t"][:, None].contiguous()] + [t["depth_s"][s][:, None].contiguous() for s in range(S)]
    pose, depth = ns["refine_window_dense"](t["tgt"], [t["srcs"][s] for s in range(S)], depths, t["K"], t["pose"], options, config, gn_iters=3)
    o = default_opts(n_iters=3, w_dc=0.15, prior_init=0.1, min_depth=0.06, max_depth=2.67, window_rule=_lib.WINDOW_REFERENCE,
                     depth_param=_lib.DEPTH_QUARTER)         # (the stub asks for the reference's quarter-resolution unknown: H, W are multiples of 4)
    rp, rd, _ = Engine(H, W, 2 * S).refine_dense_window(t["tgt"], t["srcs"], depths[0], torch.stack(depths[1:]), t["K"], t["pose"], o, argmin=True)
    torch.cuda.synchronize()
    assert torch.equal(pose, rp) and torch.equal(depth, rd[:1]) and not torch.equal(depth, depths[0])
