"""GPU parity of the JOINT dense mode (tcsfm_refine_dense_window with opts.dense_joint, the default): the S forward pairs of a
target share ONE inverse-depth map -- what the reference's optimize_depth_pred mode optimises (optimizer.py:194-198,235-247) --
and are solved together: per-pixel Schur elimination of the shared depth, ONE reduced camera system of 6S x 6S per target.

The float64 oracle (oracle/tcsfm_oracle.c linearize_joint / orc_refine_dense_joint) is pinned on the reference's autograd
gradients of its forward loss w.r.t. the poses and the SHARED target depth (golden G13, tests/test_oracle_vs_golden.py); here the
HIP kernels are checked against it with the engine's decisions replayed: poses 1e-4, per-pixel depth 1e-4 on every pixel."""
import os

import numpy as np
import pytest

import parity_util as PU

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu


def _t(a):
    return torch.as_tensor(np.ascontiguousarray(a, dtype=np.float32)).cuda()


def _eng(H, W, n):
    from tightly_coupled_sfm_amd.engine import Engine
    return Engine(H, W, n)


def _window(B, S, H, W, seed0=90, bias=True):
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import standins
    from oracle.oracle import Oracle
    w = standins.make_window(B, S, H, W, seed0=seed0)
    o64 = Oracle("f64")
    w["depth_t"] = o64.disp_to_depth(w["disp_t"], 0.06, 2.67)[1].astype(np.float32)
    w["depth_s"] = o64.disp_to_depth(w["disp_s"], 0.06, 2.67)[1].astype(np.float32)
    if bias:      # start from a target depth that is off, so that the depth block has work to do
        w["depth_t"] = (w["depth_t"] * (1 + 0.02 * np.sin(np.arange(W) / 11.0))[None, None, None, :]).astype(np.float32)
    return w


@pytest.mark.parametrize("shape,kw,argmin,rule", [
    ((1, 2, 96, 320), dict(n_iters=4), True, 0),                 # the KITTI window: target + 2 sources, min over the sources
    ((2, 2, 96, 320), dict(n_iters=4), True, 0),                 # two targets in one call
    ((2, 2, 96, 320), dict(n_iters=3), False, 0),                # no argmin: the shared depth couples the two poses (full 12 x 12)
    ((1, 2, 192, 640), dict(n_iters=4), True, 0),                # full KITTI size
    ((1, 3, 48, 160), dict(n_iters=3), True, 0),                 # three sources: 18 x 18
    ((2, 2, 96, 320), dict(n_iters=5, solver=1, lambda0=1e-3), True, 0),      # LM: all poses and the map accepted / rolled back together
], ids=["kitti-96x320", "two-targets", "no-argmin-coupled", "kitti-192x640", "three-sources", "lm"])
def test_joint_dense_vs_oracle(shape, kw, argmin, rule, oracle64):
    from oracle.oracle import default_opts as oopts
    from tightly_coupled_sfm_amd.engine import default_opts
    B, S, H, W = shape
    w = _window(B, S, H, W)
    e = _eng(H, W, 2 * S * B)
    o = default_opts(w_dc=0.0, min_depth=0.06, max_depth=2.67, **kw)
    r = PU.replay_window(e, oracle64, w, o, oopts(**kw), _t, argmin=argmin, dense=True, joint=True, rule=rule)
    nit = int(o.n_iters)
    assert np.all(r["stats"][:S * B, nit - 1, 0] < r["stats"][:S * B, 0, 0])          # the joint cost goes down
    assert not np.array_equal(r["depth"][0], w["depth_t"][0, 0])                       # and the shared map moved


def test_joint_beats_per_pair_copies_and_timing(oracle64):
    """same window, same start: the joint problem ends at a lower value of the joint cost than the per-pair-copy mode with the
    copies' inverse depths averaged afterwards (what optimizer.py did with them before); timing row for profiles/"""
    import json, time
    from oracle.oracle import default_opts as oopts
    from tightly_coupled_sfm_amd.engine import default_opts
    B, S, H, W = 1, 2, 192, 640
    w = _window(B, S, H, W)
    e = _eng(H, W, 2 * S * B)
    args = tuple(_t(w[k]) for k in ("target", "sources", "depth_t", "depth_s", "K", "first"))
    cost = lambda poses, depth: oracle64.linearize_dense_joint(w["target"][0], w["sources"][:, 0], depth, w["depth_s"][:, 0, 0], w["K"][0], poses[[0, 1]],
                                                               oopts(n_iters=1), argmin=True, w_prior=10.0, depth0=w["depth_t"][0, 0], rule=0)["cost"]
    out = {}
    for nit in (4, 8):
        res = {}
        for joint in (1, 0):
            o = default_opts(n_iters=nit, w_dc=0.0, min_depth=0.06, max_depth=2.67, dense_joint=joint)
            p, d, _ = e.refine_dense_window(*args, o, argmin=True)
            p, d = p.cpu().numpy().astype(np.float64), d.cpu().numpy()[:, 0].astype(np.float64)
            depth = d[0] if joint else 1.0 / (1.0 / d[:S * B]).mean(0)
            res[joint] = cost(p, depth)
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(50):
                e.refine_dense_window(*args, o, argmin=True)
            torch.cuda.synchronize()
            out[f"{'joint' if joint else 'copies'}_{nit}its_us_per_window"] = round((time.perf_counter() - t0) / 50 * 1e6, 1)
        out[f"joint_cost_{nit}its"] = {"start": cost(w["first"].astype(np.float64), w["depth_t"][0, 0]), "joint": res[1], "per_pair_copies_averaged": res[0]}
        assert res[1] < res[0], (nit, res)
    os.makedirs("gpurun_out", exist_ok=True)
    with open("gpurun_out/r03_joint_dense.json", "w") as f:
        json.dump({"window": f"B={B} S={S} {W}x{H}, argmin, GN", **out}, f)


def test_joint_sequence_and_lanes_bit_identical():
    """the joint mode through the batched / streamed entry points: B windows per call == one call per window, bit for bit"""
    from tightly_coupled_sfm_amd.engine import default_opts
    B, S, H, W = 3, 2, 48, 160
    w = _window(B, S, H, W)
    e = _eng(H, W, 2 * S * B)
    o = default_opts(n_iters=3, w_dc=0.0, min_depth=0.06, max_depth=2.67)
    args = tuple(_t(w[k]) for k in ("target", "sources", "depth_t", "depth_s", "K", "first"))
    p, d, _ = e.refine_dense_window(*args, o, argmin=True)
    e1 = _eng(H, W, 2 * S)
    for b in range(B):
        idx = torch.tensor([s * B + b for s in range(S)] + [S * B + s * B + b for s in range(S)], device="cuda")
        p1, d1, _ = e1.refine_dense_window(args[0][b:b + 1], args[1][:, b:b + 1].contiguous(), args[2][b:b + 1], args[3][:, b:b + 1].contiguous(), args[4][b:b + 1],
                                           args[5][idx].contiguous(), o, argmin=True)
        assert torch.equal(p1, p[idx]) and torch.equal(d1, d[idx]), b
