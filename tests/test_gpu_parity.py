"""GPU parity: the HIP engine, called through the C ABI (libtcsfm_hip.so), against
  (a) the committed golden vectors produced by the reference itself, and
  (b) the float64 CPU oracle on the same seeded inputs.

Tolerances (fp32 engine vs float64 truth):
  warp / residual maps   2e-5 absolute on [0,1] images; discrete decisions (valid / auto-mask) may differ
                         on <=0.3% of pixels (near-ties decided in fp32 vs fp64)
  cost                   1e-5 relative;  gradient / GN matrix  2e-4 of their largest entry
  refined pose           1e-4 relative (translation norm and rotation norm separately) -- BASELINE.json's bar
  refined depth scale    1e-4 relative on the depth values (|d log-scale| < 1e-4)
  refined per-pixel depth (dense mode)  1e-4 relative on every pixel

Iterate-level tests never loosen these bars: wherever the fp32 engine and the float64 oracle could take different DISCRETE
decisions (mask near-ties, min-over-sources near-ties, LM accept / reject on converged costs) the oracle replays the engine's
recorded decisions (tests/parity_util.py, tcsfm_debug_trace) and the decisions themselves are checked separately: flipped
pixels must be near-ties and few, flipped LM decisions must be cost ties.
"""
import os

import numpy as np
import pytest

from conftest import load_golden, REPO
import parity_util as PU

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu

SMALL = ["s8x16", "s24x40", "s48x160"]


def _eng(H, W, n):
    from tightly_coupled_sfm_amd.engine import Engine
    return Engine(H, W, n)


def _t(a):
    return torch.as_tensor(np.ascontiguousarray(a, dtype=np.float32)).cuda()


def _maxabs(a, b):
    return float(np.max(np.abs(np.asarray(a, np.float64) - np.asarray(b, np.float64))))


def _stack_pair(g, poses):
    n = len(poses)
    rep = lambda a: np.repeat(a[None], n, 0)
    return (_t(rep(g["tgt"])), _t(rep(g["src"])), _t(rep(g["depth_t"])[:, None]), _t(rep(g["depth_s"])[:, None]),
            _t(rep(g["K"])), _t(poses))


@pytest.mark.parametrize("name", SMALL)
def test_warp_vs_reference_golden(name):
    """inverse_warp2 (stn.py:234-273) incl. OOB sentinel, zero-pad border blending, Z clamp"""
    g = load_golden(name)
    H, W = g["tgt"].shape[1:]
    tgt, src, dt, ds, K, poses = _stack_pair(g, g["poses"])
    e = _eng(H, W, len(poses))
    rec, valid, pd, cd = e.inverse_warp2(src, dt, ds, -poses, K)   # reference call form: -pose (train_mono.py:69)
    rec, valid, pd, cd = (x.cpu().numpy() for x in (rec, valid, pd, cd))
    for k in range(len(poses)):
        bad = valid[k, 0] != g["f64_valid"][k]
        assert bad.mean() <= 0.003 or (not bad[1:-1, 1:-1].any()), (name, k, int(bad.sum()))
        ok = ~bad
        assert _maxabs(rec[k][:, ok], g["f64_rec"][k][:, ok]) < 1e-4      # |ix| up to W: ulp(ix)*gradient
        assert _maxabs(pd[k, 0][ok], g["f64_proj_depth"][k][ok]) < 2e-4
        scale = max(1.0, float(np.abs(g["f64_comp_depth"][k]).max()))
        assert _maxabs(cd[k, 0], g["f64_comp_depth"][k]) < 1e-5 * scale


@pytest.mark.parametrize("name", SMALL)
def test_photometric_vs_reference_golden(name):
    """compute_photometric_error (helpers.py:8-23): diff, weight, combined mask"""
    g = load_golden(name)
    H, W = g["tgt"].shape[1:]
    tgt, src, dt, ds, K, poses = _stack_pair(g, g["poses"])
    e = _eng(H, W, len(poses))
    r = e.compute_photometric_error(tgt, src, dt, ds, poses, K)
    diff, mask, weight, wv = (r[k].cpu().numpy()[:, 0] for k in ("diff_img", "valid_mask", "weight_mask", "warp_valid"))
    for k in range(len(poses)):
        vbad = wv[k] != g["f64_valid"][k]
        if vbad.any():      # exact-tie border pixels at the identity pose / a handful of fp32 ties
            assert vbad.mean() <= 0.003 or not vbad[1:-1, 1:-1].any()
            continue
        assert _maxabs(diff[k], g["f64_diff"][k]) < 3e-5
        assert _maxabs(weight[k], g["f64_weight"][k]) < 1e-4
        assert (mask[k] != g["f64_mask"][k]).mean() <= 0.003


def test_batched_fwd_inv_stack_vs_solve_pose_iteratively():
    """the [fwd ; inv] directed-pair stacking of solve_pose_iteratively (train_mono.py:54-62,82-100) with a
    constant-pose PoseNet stand-in, iterations=1: error images of all 8 directed pairs in ONE engine call"""
    g = load_golden("batch24x40")
    B, S = 2, 2
    target, sources, depths, K = g["target"], g["sources"], g["depths"], g["K"]
    H, W = target.shape[2:]
    tg = np.concatenate([target] * S); sr = np.concatenate(list(sources))
    dtg = np.concatenate([depths[0]] * S); dsr = np.concatenate(list(depths[1:]))
    tgt = np.concatenate([tg, sr]); src = np.concatenate([sr, tg])
    dt = np.concatenate([dtg, dsr]); ds = np.concatenate([dsr, dtg])
    Ks = np.concatenate([K] * (2 * S))
    e = _eng(H, W, 2 * S * B)
    r = e.compute_photometric_error(_t(tgt), _t(src), _t(dt), _t(ds), _t(g["first"]), _t(Ks))
    split = S * B
    for d, sl in (("fwd", slice(0, split)), ("inv", slice(split, None))):
        assert _maxabs(r["diff_img"].cpu().numpy()[sl], g[f"it1_{d}_diff_img"]) < 3e-5
        assert _maxabs(r["weight_mask"].cpu().numpy()[sl], g[f"it1_{d}_weight_mask"]) < 1e-4
        assert _maxabs(r["auto_mask_error"].cpu().numpy()[sl], g[f"it1_{d}_auto_mask_error"]) < 3e-5
        assert (r["warp_valid"].cpu().numpy()[sl] != g[f"it1_{d}_valid_mask"]).mean() <= 0.003
        assert (r["auto_mask"].cpu().numpy()[sl] != g[f"it1_{d}_auto_mask"]).mean() <= 0.003
        assert _maxabs(r["img_rec"].cpu().numpy()[sl], g[f"it1_{d}_img_rec"]) < 1e-4


def test_optimization_loss_vs_reference_golden():
    """compute_optimization_loss (optimizer.py:29-134) under the default options and each toggle (golden G5), assembled
    from maps the HIP library produced for the stacked fwd/inv batch; SSIM_Loss standalone (losses.py:27-41)"""
    from tightly_coupled_sfm_amd.losses import compute_optimization_loss
    g = load_golden("batch24x40")
    B, S = 2, 2
    target, sources, depths, K = g["target"], g["sources"], g["depths"], g["K"]
    H, W = target.shape[2:]
    tg = np.concatenate([target] * S); sr = np.concatenate(list(sources))
    dtg = np.concatenate([depths[0]] * S); dsr = np.concatenate(list(depths[1:]))
    e = _eng(H, W, 2 * S * B)
    split = S * B
    for iters in (1, 4):
        poses = np.concatenate([g[f"it{iters}_fwd_poses"][:, -1], g[f"it{iters}_inv_poses"][:, -1]])
        r = e.compute_photometric_error(_t(np.concatenate([tg, sr])), _t(np.concatenate([sr, tg])), _t(np.concatenate([dtg, dsr])),
                                        _t(np.concatenate([dsr, dtg])), _t(poses), _t(np.concatenate([K] * (2 * S))))
        def part(sl, stacked):
            return {"diff_img": r["diff_img"][sl], "valid_mask": r["warp_valid"][sl], "weight_mask": r["weight_mask"][sl],
                    "auto_mask_error": r["auto_mask_error"][sl], "auto_mask": r["auto_mask"][sl], "poses": _t(stacked)}
        fwd = part(slice(0, split), g[f"it{iters}_fwd_poses"]); inv = part(slice(split, None), g[f"it{iters}_inv_poses"])
        base = {'diff_img_argmin': True, 'automasking': True, 'l_depth_consist': True, 'l_depth_consist_weight': 0.15,
                'l_depth_init': True, 'l_depth_init_weight': 0.1, 'l_inverse_reconstruction': True, 'l_smooth': False,
                'l_smooth_weight': 2, 'l_pose_consist': False, 'num_source_imgs': S}
        for tag, upd in (("default", {}), ("noargmin", {'diff_img_argmin': False}), ("noauto", {'automasking': False}),
                         ("noinv", {'l_inverse_reconstruction': False}), ("nodc", {'l_depth_consist': False}),
                         ("smooth", {'l_smooth': True}), ("posec", {'l_pose_consist': True}), ("noinit", {'l_depth_init': False})):
            loss = compute_optimization_loss(dict(base, **upd), _t(target), _t(g["loss_disp"]), _t(g["loss_disp0"]), fwd, inv, e.ssim_loss)
            ref = float(g[f"it{iters}_loss_{tag}"])
            assert abs(float(loss.reshape(-1)[0]) - ref) < 2e-4 * abs(ref), (iters, tag, float(loss.reshape(-1)[0]), ref)
    s = e.ssim_loss(_t(target), _t(sources[0])).cpu().numpy()
    import sys
    from oracle.oracle import Oracle
    so = np.stack([Oracle("f64").ssim(target[b], sources[0][b]) for b in range(B)])
    assert _maxabs(s, so) < 2e-6


def test_smooth_loss_kernel_vs_reference_expression():
    """get_smooth_loss (losses.py:43-61) as a HIP kernel against the reference's expression evaluated in float64 torch, and the
    drop-in losses.get_smooth_loss routing GPU tensors through it (the expression itself is pinned on golden G5's `smooth` toggle)"""
    from tightly_coupled_sfm_amd import losses
    g = load_golden("batch24x40")
    disp, img = g["loss_disp"], g["target"]
    H, W = img.shape[2:]
    d64, i64 = torch.tensor(disp, dtype=torch.float64), torch.tensor(img, dtype=torch.float64)
    ref = float(losses.get_smooth_loss(d64, i64))
    e = _eng(H, W, disp.shape[0])
    got = e.smooth_loss(_t(disp), _t(img))
    assert abs(got - ref) < 2e-6 * abs(ref), (got, ref)
    assert abs(float(losses.get_smooth_loss(_t(disp), _t(img))) - ref) < 2e-6 * abs(ref)
    rng = np.random.default_rng(5)                       # a size with ragged blocks, non-trivial values
    e2 = _eng(37, 53, 3)
    dd, ii = rng.uniform(0.05, 0.9, (3, 1, 37, 53)), rng.uniform(0, 1, (3, 3, 37, 53))
    r2 = float(losses.get_smooth_loss(torch.tensor(dd), torch.tensor(ii)))
    assert abs(e2.smooth_loss(_t(dd), _t(ii)) - r2) < 2e-6 * r2


def test_loss_surface_vs_reference_golden():
    """generate_loss_surface (plot_loss_surface.py:11-87): both 50-point sweeps in one launch each"""
    g = load_golden("sweep48x160")
    H, W = g["tgt"].shape[1:]
    e = _eng(H, W, 64)
    args = (_t(g["tgt"][None]), _t(g["src"][None]), _t(g["depth_t"][None, None]), _t(g["depth_s"][None, None]), _t(g["K"][None]))
    for deltas, errs, idx in ((g["delta_list"], g["errors"], 2), (g["delta_list_yaw"], g["errors_yaw"], 4)):
        poses = np.repeat(g["pose"][None].astype(np.float32), len(deltas), 0)
        poses[:, idx] += deltas
        mine = e.loss_surface(*args, _t(poses))
        rel = np.abs(mine - errs) / errs
        assert np.median(rel) < 2e-5 and rel.max() < 2e-3
        assert abs(int(np.argmin(mine)) - int(np.argmin(errs))) <= 1


def test_disp_to_depth_vs_reference_golden():
    g = load_golden("helpers")
    e = _eng(8, 8, 1)
    s, d = e.disp_to_depth(_t(g["disp"]), 0.06, 2.67)
    assert _maxabs(s.cpu().numpy(), g["scaled_disp"]) < 2e-6 * g["scaled_disp"].max()
    assert np.max(np.abs(d.cpu().numpy() - g["depth"]) / g["depth"]) < 2e-6


def _pairs(N, H, W, seed0=0, both=False):
    from tightly_coupled_sfm_amd import synth
    return synth.make_batch(N, H, W, seed0=seed0, both_directions=both)


def _dev(b):
    return (_t(b["tgt"]), _t(b["src"]), _t(b["depth_t"]), _t(b["depth_s"]), _t(b["K"]))


@pytest.mark.parametrize("H,W", [(24, 40), (48, 160), (192, 640)])
@pytest.mark.parametrize("refine,w_dc", [(0, 0.0), (0, 0.15), (1, 0.0), (1, 0.15)])
def test_linearize_vs_oracle(H, W, refine, w_dc, oracle64):
    """cost, mask count, exact gradient and GN matrix of one linearisation vs the float64 oracle"""
    from oracle.oracle import default_opts as oopts
    from tightly_coupled_sfm_amd.engine import default_opts
    N = 3
    b = _pairs(N, H, W, seed0=4)
    e = _eng(H, W, N)
    ls = np.array([0.0, 0.03, -0.05], dtype=np.float32)
    o = default_opts(refine=refine, w_dc=w_dc)
    out = e.linearize(*_dev(b), _t(b["pose_init"]), o, log_scale=_t(ls) if refine else None)
    for n in range(N):
        ref = oracle64.linearize(b["tgt"][n], b["src"][n], b["depth_t"][n, 0], b["depth_s"][n, 0], b["pose_init"][n], b["K"][n],
                                 oopts(nparam=6 + refine, w_dc=w_dc), log_scale=float(ls[n]) if refine else 0.0)
        assert abs(out["cost"][n] - ref["cost"]) < 1e-5 * ref["cost"]
        assert abs(out["n_mask"][n] - ref["n_mask"]) <= 0.003 * ref["n_mask"] + 1
        assert _maxabs(out["g"][n], ref["g"]) < 2e-4 * np.abs(ref["g"]).max()
        assert _maxabs(out["H"][n], ref["H"]) < 2e-4 * np.abs(ref["H"]).max()


CASES = [dict(), dict(solver=1, lambda0=1e-3, n_iters=4), dict(param=1), dict(w_dc=0.15), dict(refine=1), dict(refine=1, w_dc=0.15),
         dict(automask=0), dict(n_iters=1), dict(n_iters=8)]


@pytest.mark.parametrize("kw", CASES, ids=[",".join(f"{k}={v}" for k, v in c.items()) or "default" for c in CASES])
def test_refine_vs_oracle(kw, oracle64):
    """N iterations of GN / LM: refined poses within 1e-4 relative of the float64 CPU twin (BASELINE.json bar)"""
    from oracle.oracle import default_opts as oopts
    from tightly_coupled_sfm_amd.engine import default_opts
    H, W, N = 96, 320, 4
    b = _pairs(N, H, W, seed0=10, both=True)
    e = _eng(H, W, N)
    o = default_opts(**kw)
    ls0 = np.array([0.02, -0.02, 0.0, 0.05], dtype=np.float32)
    refine = kw.get("refine", 0)
    pose, ls, st = e.refine(*_dev(b), _t(b["pose_init"]), o, log_scale=_t(ls0) if refine else None, stats=True)
    pose, st = pose.cpu().numpy().astype(np.float64), st.cpu().numpy()
    okw = dict(kw); okw["nparam"] = 6 + okw.pop("refine", 0)
    for n in range(N):
        rp, rls, rst = oracle64.refine(b["tgt"][n], b["src"][n], b["depth_t"][n, 0], b["depth_s"][n, 0], b["pose_init"][n], b["K"][n],
                                       oopts(**okw), log_scale=float(ls0[n]) if refine else 0.0)
        et = np.linalg.norm(pose[n, :3] - rp[:3]) / np.linalg.norm(rp[:3])
        er = np.linalg.norm(pose[n, 3:] - rp[3:]) / np.linalg.norm(rp[3:])
        assert et < 1e-4 and er < 1e-4, (n, et, er, pose[n], rp)
        if refine:   # depth = exp(log_scale) * depth0: 1e-4 relative on every depth value
            assert abs(float(ls[n]) - rls) < 1e-4
        nrow = rst.shape[0] if o.solver == 1 else rst.shape[0] - 1
        assert np.max(np.abs(st[n, :nrow, 0] - rst[:nrow, 0]) / rst[:nrow, 0]) < 2e-5   # cost trajectory
        assert np.all(pose[n] != b["pose_init"][n].astype(np.float64))                 # something moved


def test_full_size_properties():
    """BASELINE size (640x192): properties that need no oracle run"""
    from tightly_coupled_sfm_amd.engine import default_opts
    H, W, N = 192, 640, 6
    b = _pairs(N, H, W, seed0=0, both=True)
    e = _eng(H, W, N)
    d = _dev(b)
    p0 = _t(b["pose_init"])
    # 0 iterations is the identity on the pose (round trip pose -> SE(3) -> pose)
    pz, _, _ = e.refine(*d, p0, default_opts(n_iters=0))
    assert _maxabs(pz.cpu().numpy(), b["pose_init"]) < 2e-7
    # monotone cost under LM, and convergence towards the ground truth under GN
    pl, _, st = e.refine(*d, p0, default_opts(solver=1, n_iters=8, lambda0=1e-3), stats=True)
    st = st.cpu().numpy()
    assert np.all(st[:, -1, 0] <= st[:, 0, 0])
    pg, _, sg = e.refine(*d, p0, default_opts(n_iters=8), stats=True)
    sg = sg.cpu().numpy()
    assert np.all(sg[:, 7, 0] < 0.8 * sg[:, 0, 0])
    # the stats rows carry the trajectory of iterates: row 0 = initial pose, row n_iters = refined pose
    assert _maxabs(sg[:, 0, 4:10], b["pose_init"]) < 2e-7 and np.array_equal(sg[:, 8, 4:10], pg.cpu().numpy())
    assert np.all(np.abs(np.diff(sg[:, :, 4:10], axis=1)).max(axis=2) > 0)
    # determinism + batch independence: a pair refined alone gives bit-identical results to the same pair in a batch
    pg2, _, _ = e.refine(*d, p0, default_opts(n_iters=8))
    assert torch.equal(pg, pg2)
    one = tuple(x[2:3].contiguous() for x in d)
    ps, _, _ = e.refine(*one, p0[2:3].contiguous(), default_opts(n_iters=8))
    assert torch.equal(ps[0], pg[2])
    # permutation of the batch permutes the result
    perm = torch.tensor([3, 0, 5, 1, 4, 2], device=p0.device)
    pp, _, _ = e.refine(*(x[perm].contiguous() for x in d), p0[perm].contiguous(), default_opts(n_iters=8))
    assert torch.equal(pp, pg[perm])


def test_edge_cases():
    from tightly_coupled_sfm_amd.engine import default_opts
    H, W = 24, 40
    b = _pairs(2, H, W, seed0=1)
    e = _eng(H, W, 2)
    d = _dev(b)
    # pose that throws every pixel out of bounds: nothing is valid, the pose must come back unchanged and finite
    far = _t(np.array([[0, 0, 0, 0, 1.5, 0], [0, 0, 0, 1.5, 0, 0]], dtype=np.float32))
    p, _, st = e.refine(*d, far, default_opts(n_iters=3), stats=True)
    assert torch.isfinite(p).all() and _maxabs(p.cpu().numpy(), far.cpu().numpy()) < 1e-6
    assert float(st[:, 0, 2].abs().max()) == 0.0
    # non-pinhole intrinsics are refused with an error, not silently mis-handled
    Kbad = d[4].clone(); Kbad[:, 0, 1] = 0.5
    with pytest.raises(RuntimeError, match="pinhole"):
        e.refine(d[0], d[1], d[2], d[3], Kbad, _t(b["pose_init"]))
    # more pairs than the handle was created for
    b3 = _pairs(3, H, W)
    with pytest.raises(RuntimeError, match="out of range"):
        e.refine(*_dev(b3), _t(b3["pose_init"]))
    # reference-style size check (stn.py:24-30)
    with pytest.raises(AssertionError, match="wrong size"):
        e.refine(d[0][:, :, :-1], d[1], d[2], d[3], d[4], _t(b["pose_init"]))
    # sigmoid-disparity inputs with the fused disp_to_depth equal explicit depths
    from tightly_coupled_sfm_amd import synth
    sd_t, sd_s = _t(synth.depth_to_sigmoid_disp(b["depth_t"].astype(np.float64))), _t(synth.depth_to_sigmoid_disp(b["depth_s"].astype(np.float64)))
    pa, _, _ = e.refine(*d, _t(b["pose_init"]), default_opts())
    pb, _, _ = e.refine(d[0], d[1], sd_t, sd_s, d[4], _t(b["pose_init"]), default_opts(depth_is_disp=1, min_depth=0.06, max_depth=2.67))
    assert _maxabs(pa.cpu().numpy(), pb.cpu().numpy()) < 5e-5


def _perturbed_depth(b):
    H, W = b["depth_t"].shape[2:]
    v, u = np.mgrid[0:H, 0:W]
    return (b["depth_t"] * (1 + 0.03 * np.sin(u / 23.0) * np.cos(v / 17.0))[None, None]).astype(np.float32)


@pytest.mark.parametrize("H,W", [(24, 40), (96, 320)])
def test_dense_refine_vs_oracle(H, W, oracle64):
    """dense mode (BASELINE config 5 shape of problem): pose + per-pixel inverse depth with per-pixel Schur elimination.
    Pose within 1e-4 relative and per-pixel depth within 1e-4 relative (mask decisions replayed; parity_util.assert_depth)."""
    from oracle.oracle import default_opts as oopts
    from tightly_coupled_sfm_amd.engine import default_opts
    N = 2
    b = _pairs(N, H, W, seed0=3, both=True)
    d0 = _perturbed_depth(b)
    o = default_opts(n_iters=4, lambda_depth=1.0, prior_depth=10.0)
    r = PU.replay_dense_pairs(_eng(H, W, N), oracle64, b, d0, o, oopts(n_iters=4), _t)
    st, depth = r["stats"], r["depth"]
    for n in range(N):
        assert st[n, 3, 0] < st[n, 0, 0]                      # the joint refinement lowers the cost
        assert np.abs(depth[n] / d0[n, 0] - 1).max() > 1e-3   # and the depth map really moved
        # the free-running oracle agrees on all but the tie pixels
        rp, rd, _ = oracle64.refine_dense(b["tgt"][n], b["src"][n], d0[n, 0], b["depth_s"][n, 0], b["pose_init"][n], b["K"][n],
                                          oopts(n_iters=4), lambda_depth=1.0, w_prior=10.0)
        assert np.mean(np.abs(depth[n] / rd - 1) < 1e-4) >= 0.998


def test_dense_pose_block_equals_pose_mode(oracle64):
    """with the depth block frozen (huge lambda_depth, no prior) one dense iteration takes exactly the pose-mode step:
    the adjoint-form kernel and the forward-form kernel implement the same gradient and curvature"""
    from tightly_coupled_sfm_amd.engine import default_opts
    H, W, N = 48, 160, 2
    b = _pairs(N, H, W, seed0=6, both=True)
    e = _eng(H, W, N)
    d = _dev(b)
    p0 = _t(b["pose_init"])
    pa, _, _ = e.refine(*d, p0, default_opts(n_iters=2))
    pb, depth, _ = e.refine_dense(d[0], d[1], d[2], d[3], d[4], p0, default_opts(n_iters=2, lambda_depth=1e30, prior_depth=0.0))
    ra = pa.cpu().numpy(); rb = pb.cpu().numpy()
    assert np.max(np.abs(ra - rb)) < 2e-6 * np.abs(ra).max()
    assert _maxabs(depth.cpu().numpy(), b["depth_t"]) < 1e-6


def test_scale_recovery_vs_reference_golden(oracle64):
    """DNet ScaleRecovery (dnet_layers.py:249-327): height map, ground mask, exact batch median and scale against the
    reference's own run (golden G10) -- the median by an integer-histogram radix select, no sort"""
    g = load_golden("scale48x160")
    B, H, W = g["depth"].shape
    e = _eng(H, W, B)
    scale, med, hm, mm = e.scale_recovery(_t(g["depth"][:, None]), _t(g["K"]), float(g["cam_height"]), maps=True)
    assert (mm.cpu().numpy()[:, 0] != g["f64_mask"]).mean() <= 0.002
    ok = mm.cpu().numpy()[:, 0] == g["f64_mask"]
    # fp32 normals from 1-pixel depth differences: ~1e-5 relative on the height (the reference's own fp32 run: same order)
    assert _maxabs(hm.cpu().numpy()[:, 0][ok], g["f64_height"][ok]) < 2e-5 * float(g["f64_height"].max())
    assert abs(float(med) - float(g["f32_median"])) <= 2e-7 and abs(float(scale) - float(g["f32_scale"][0])) < 1e-5
    # batch padding with copies of image 0, as the reference does for a short batch
    one = e.scale_recovery(_t(g["depth"][:1, None]), _t(g["K"][:1]), float(g["cam_height"]), pad_to_batch=4)
    s1, m1 = oracle64.scale_recovery(np.repeat(g["depth"][:1], 4, 0), np.repeat(g["K"][:1], 4, 0), float(g["cam_height"]))
    assert abs(float(one) - s1) < 1e-5 * s1
    # ... and against the reference itself: its batch-size-5 layer fed with this batch of 2
    pad5 = e.scale_recovery(_t(g["depth"][:, None]), _t(g["K"]), float(g["cam_height"]), pad_to_batch=5)
    assert abs(float(pad5) - float(g["f32_scale_pad5"][0])) < 1e-5 * float(g["f32_scale_pad5"][0])
    assert abs(float(g["f32_scale_pad5"][0]) - float(g["f32_scale"][0])) > 0.05           # the padding does change the median
    # no ground at all -> NaN, not garbage
    up = _t(np.full((1, 1, H, W), 0.5, dtype=np.float32))
    assert torch.isnan(_eng(H, W, 1).scale_recovery(up, _t(g["K"][:1]), 0.055))


def test_posenet_input_assembly_vs_reference_golden():
    """outputs['comb']['imgs'] of solve_pose_iteratively (train_mono.py:104-107): (tgt * valid | img_rec), produced by the warp"""
    g = load_golden("batch24x40")
    B, S = 2, 2
    target, sources, depths, K = g["target"], g["sources"], g["depths"], g["K"]
    H, W = target.shape[2:]
    tg = np.concatenate([target] * S); sr = np.concatenate(list(sources))
    dtg = np.concatenate([depths[0]] * S); dsr = np.concatenate(list(depths[1:]))
    e = _eng(H, W, 2 * S * B)
    out = e.posenet_input(_t(np.concatenate([tg, sr])), _t(np.concatenate([sr, tg])), _t(np.concatenate([dtg, dsr])),
                          _t(np.concatenate([dsr, dtg])), _t(g["first"]), _t(np.concatenate([K] * (2 * S)))).cpu().numpy()
    rec = np.concatenate([g["it1_fwd_img_rec"], g["it1_inv_img_rec"]])
    valid = np.concatenate([g["it1_fwd_valid_mask"], g["it1_inv_valid_mask"]])
    tgt_all = np.concatenate([tg, sr])
    assert _maxabs(out[:, 3:6], rec) < 1e-4
    assert np.mean(np.abs(out[:, 0:3] - tgt_all * valid) > 1e-6) <= 0.003


@pytest.mark.parametrize("H,W", [(240, 320), (256, 448)], ids=["320x240 (BASELINE config 5)", "448x256 (the reference's ScanNet size)"])
def test_dense_refine_scannet_size(H, W, oracle64):
    """BASELINE config 5 at both sizes SURVEY 8d names: 320x240 as BASELINE.json states, and 448x256, the reference's own ScanNet
    resolution (scannet_test_loader.py:23-24); depth range of run_scannet_exps.sh:3; fwd + inv directed pairs, 4 GN iterations"""
    from oracle.oracle import default_opts as oopts
    from tightly_coupled_sfm_amd.engine import default_opts
    b = _pairs(2, H, W, seed0=8, both=True)
    d0 = _perturbed_depth(b)
    o = default_opts(n_iters=4, lambda_depth=1.0, prior_depth=10.0, min_depth=0.03, max_depth=3.0)
    r = PU.replay_dense_pairs(_eng(H, W, 2), oracle64, b, d0, o, oopts(n_iters=4), _t)
    assert np.all(r["stats"][:, 3, 0] < r["stats"][:, 0, 0])


@pytest.mark.parametrize("H,W", [(96, 320), (192, 640)])
def test_pose_scale_8_iters_config4(H, W, oracle64):
    """BASELINE config 4: depth-scale + 6-DoF joint refinement, 8 iterations (Eigen-split depth range 0.1 .. 2.67), also at the
    full 640x192: free-running oracle AND decision replay"""
    from oracle.oracle import default_opts as oopts
    from tightly_coupled_sfm_amd.engine import default_opts
    N = 2
    b = _pairs(N, H, W, seed0=30, both=(W == 640))
    e = _eng(H, W, N)
    ls0 = np.array([0.04, -0.03], dtype=np.float32)
    r = PU.replay_pairs(e, oracle64, b, default_opts(refine=1, n_iters=8), oopts(nparam=7, n_iters=8), _t, log_scale=ls0)
    for n in range(N):    # and against the oracle's own decisions
        rp, rls, _ = oracle64.refine(b["tgt"][n], b["src"][n], b["depth_t"][n, 0], b["depth_s"][n, 0], b["pose_init"][n], b["K"][n],
                                     oopts(nparam=7, n_iters=8), log_scale=float(ls0[n]))
        PU.assert_pose(r["pose"][n], rp, ("free", n))
        assert abs(float(r["log_scale"][n]) - rls) < 1e-4


# ------------------------------------------------------------------------------------------------------------------
# window form + per-pixel min over the sources (compute_optimization_loss, optimizer.py:47-69)
def _window(B, S, H, W):
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import standins
    from oracle.oracle import Oracle
    w = standins.make_window(B, S, H, W, seed0=90)
    o64 = Oracle("f64")
    w["depth_t"] = o64.disp_to_depth(w["disp_t"], 0.06, 2.67)[1].astype(np.float32)
    w["depth_s"] = o64.disp_to_depth(w["disp_s"], 0.06, 2.67)[1].astype(np.float32)
    return w


@pytest.mark.parametrize("kw", [dict(), dict(solver=1, n_iters=4, lambda0=1e-3), dict(refine=1, n_iters=3)],
                         ids=["gn", "lm", "pose+scale"])
def test_refine_window_argmin_vs_oracle(oracle64, kw):
    """window form with the per-pixel min over the sources (optimizer.py:47-69), GN / LM / pose+scale: 1e-4 on every pair with the
    engine's selection and LM decisions replayed; the decisions themselves bounded (parity_util)"""
    from oracle.oracle import default_opts as oopts
    from tightly_coupled_sfm_amd.engine import default_opts
    B, S, H, W = 2, 2, 96, 320
    w = _window(B, S, H, W)
    e = _eng(H, W, 2 * S * B)
    okw = dict(kw); okw["nparam"] = 6 + okw.pop("refine", 0)
    r = PU.replay_window(e, oracle64, w, default_opts(**kw), oopts(**okw), _t, argmin=True,
                         log_scale=np.zeros(2 * S * B, np.float32) if kw.get("refine") else None)
    # the forward pairs of one target partition the kept pixels; both sources win somewhere
    assert np.all(r["stats"][:S * B, 0, 2] > 0.02 * H * W)
    # and the free-running oracle (its own decisions) sees the same masks up to a handful of tie pixels
    rp, _, rst = oracle64.refine_window(w["target"], w["sources"], w["depth_t"][:, 0], w["depth_s"][:, :, 0], w["K"], w["first"],
                                        oopts(**okw), argmin=True, log_scale=np.zeros(2 * S * B) if kw.get("refine") else None)
    nlin = int(default_opts(**kw).n_iters)
    assert np.max(np.abs(r["stats"][:, :nlin, 2] - rst[:, :nlin, 2])) <= 4


def test_refine_window_without_argmin_is_the_pair_form():
    from tightly_coupled_sfm_amd.engine import default_opts
    B, S, H, W = 2, 2, 96, 320
    w = _window(B, S, H, W)
    e = _eng(H, W, 2 * S * B)
    tg, sr, dt, ds, K, p0 = (_t(w[k]) for k in ("target", "sources", "depth_t", "depth_s", "K", "first"))
    pw, _, sw = e.refine_window(tg, sr, dt, ds, K, p0, default_opts(n_iters=4), stats=True, argmin=False)
    # the stacked tensors solve_pose_iteratively builds (train_mono.py:54-62)
    T = tg.repeat(S, 1, 1, 1); Sx = sr.reshape(S * B, 3, H, W)
    Dt = dt.repeat(S, 1, 1, 1); Ds = ds.reshape(S * B, 1, H, W)
    pp, _, sp = e.refine(torch.cat([T, Sx]), torch.cat([Sx, T]), torch.cat([Dt, Ds]), torch.cat([Ds, Dt]), K.repeat(2 * S, 1, 1), p0,
                         default_opts(n_iters=4), stats=True)
    assert torch.equal(pw, pp) and torch.equal(sw, sp)
    # one source: argmin has nothing to choose from
    p1, _, _ = e.refine_window(tg, sr[:1], dt, ds[:1], K, torch.cat([p0[:B], p0[S * B:S * B + B]]), default_opts(n_iters=4), argmin=True)
    assert torch.equal(p1[:B], pw[:B]) and torch.equal(p1[B:], pw[S * B:S * B + B])
    with pytest.raises(RuntimeError, match="max_pairs"):
        _eng(H, W, 4).refine_window(tg, sr, dt, ds, K, p0)


def test_engine_follows_torch_stream():
    """tensor-level calls run on torch's CURRENT stream (also inside a `with torch.cuda.stream(...)` block), so producers
    and consumers of the tensors stay ordered without explicit synchronisation"""
    from tightly_coupled_sfm_amd.engine import default_opts
    H, W = 96, 320
    b = _pairs(2, H, W, seed0=3)
    e = _eng(H, W, 2)
    d = _dev(b); p0 = _t(b["pose_init"])
    ref, _, _ = e.refine(*d, p0, default_opts(n_iters=3))
    torch.cuda.synchronize()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        scale = torch.full((1,), 2.0, device="cuda")
        tgt2 = d[0] * scale / scale                       # produced on the side stream right before the call
        out, _, _ = e.refine(tgt2, *d[1:], p0, default_opts(n_iters=3))
        moved = out + 0.0                                 # consumed on the side stream right after
    side.synchronize()
    assert torch.equal(moved, ref)
    g = torch.cuda.CUDAGraph()                            # ... which also makes a refine call capturable in a HIP graph
    with torch.cuda.stream(side):
        e.refine(*d, p0, default_opts(n_iters=3))
    torch.cuda.synchronize()
    buf = torch.empty_like(p0)
    with torch.cuda.graph(g, stream=side):
        e._bind()
        e.refine_into(*d, p0, buf, default_opts(n_iters=3))
    buf.zero_()
    g.replay()
    torch.cuda.synchronize()
    assert torch.equal(buf, ref)


def test_large_frame_uses_group_reduction_and_matches_oracle(oracle64):
    """KITTI-raw sized frame (375x1242: ragged tiles, 936 workgroups per pair) -- above 256 workgroups the kernel keeps
    the in-launch two-level reduction (tickets), which the 640x192 cases no longer exercise"""
    from oracle.oracle import default_opts as oopts
    from tightly_coupled_sfm_amd.engine import default_opts
    H, W, N = 375, 1242, 2
    b = _pairs(N, H, W, seed0=21, both=True)
    e = _eng(H, W, N)
    d = _dev(b); p0 = _t(b["pose_init"])
    lin = e.linearize(*d, p0)
    pose, _, st = e.refine(*d, p0, default_opts(n_iters=3), stats=True)
    pose2, _, _ = e.refine(*d, p0, default_opts(n_iters=3))
    assert torch.equal(pose, pose2)                                   # deterministic through the ticket path too
    for n in range(N):
        r = oracle64.linearize(b["tgt"][n], b["src"][n], b["depth_t"][n, 0], b["depth_s"][n, 0], b["pose_init"][n], b["K"][n], oopts())
        assert abs(lin["n_mask"][n] - r["n_mask"]) <= 0.003 * r["n_mask"] + 1 and abs(lin["cost"][n] - r["cost"]) < 1e-5 * r["cost"]
        assert _maxabs(lin["g"][n], r["g"]) < 2e-4 * np.abs(r["g"]).max()
        assert _maxabs(lin["H"][n], r["H"]) < 2e-4 * np.abs(r["H"]).max()
    rr = PU.replay_pairs(e, oracle64, b, default_opts(n_iters=3), oopts(n_iters=3), _t)
    assert np.array_equal(rr["pose"].astype(np.float32), pose.cpu().numpy())      # the trace does not change the result


def test_host_pointer_calls_match_device_pointer_calls():
    """opts.host_ptrs = 1: the library stages host arrays itself (what a C caller without device memory uses);
    results are bit-identical to the device-pointer path"""
    import ctypes as C
    from tightly_coupled_sfm_amd import _lib
    from tightly_coupled_sfm_amd.engine import default_opts
    H, W, N = 48, 160, 2
    b = _pairs(N, H, W, seed0=8, both=True)
    e = _eng(H, W, N)
    d = _dev(b); p0 = _t(b["pose_init"])
    ref_pose, _, ref_st = e.refine(*d, p0, default_opts(n_iters=3), stats=True)
    ref_rec, ref_valid, ref_pd, ref_cd = e.inverse_warp2(d[1], d[2], d[3], -p0, d[4])
    torch.cuda.synchronize()
    host = {k: np.ascontiguousarray(b[k], dtype=np.float32) for k in ("tgt", "src", "depth_t", "depth_s", "K", "pose_init")}
    ptr = lambda a: a.ctypes.data_as(C.c_void_p)
    o = default_opts(n_iters=3, host_ptrs=1)
    pose = np.zeros((N, 6), np.float32); st = np.zeros((N, 4, _lib.NSTAT), np.float32)
    e._call(e.lib.tcsfm_refine(e._h, C.byref(o), N, ptr(host["tgt"]), ptr(host["src"]), ptr(host["depth_t"]), ptr(host["depth_s"]),
                               ptr(host["K"]), ptr(host["pose_init"]), None, ptr(pose), None, ptr(st)))
    assert np.array_equal(pose, ref_pose.cpu().numpy()) and np.array_equal(st, ref_st.cpu().numpy())
    rec = np.zeros((N, 3, H, W), np.float32); valid, pd, cd = (np.zeros((N, 1, H, W), np.float32) for _ in range(3))
    e._call(e.lib.tcsfm_warp(e._h, C.byref(o), N, ptr(host["src"]), ptr(host["depth_t"]), ptr(host["depth_s"]), ptr(host["pose_init"]),
                             ptr(host["K"]), ptr(rec), ptr(valid), ptr(pd), ptr(cd)))
    assert np.array_equal(rec, ref_rec.cpu().numpy()) and np.array_equal(valid, ref_valid.cpu().numpy())
    assert np.array_equal(pd, ref_pd.cpu().numpy()) and np.array_equal(cd, ref_cd.cpu().numpy())
    # window form from host memory: B=1, S=2 (fwd pairs then inv pairs)
    tg, sr = host["tgt"][:1], np.stack([host["src"][:1], host["tgt"][1:2]])            # two "sources" for target 0
    dt, ds = host["depth_t"][:1], np.stack([host["depth_s"][:1], host["depth_t"][1:2]])
    p4 = np.ascontiguousarray(np.concatenate([host["pose_init"][:1], host["pose_init"][:1], host["pose_init"][1:2], host["pose_init"][1:2]]))
    e4 = _eng(H, W, 4)
    out_h = np.zeros((4, 6), np.float32)
    ow = default_opts(n_iters=2, host_ptrs=1, argmin=1)
    e4._call(e4.lib.tcsfm_refine_window(e4._h, C.byref(ow), 1, 2, ptr(tg), ptr(sr), ptr(dt), ptr(ds), ptr(host["K"][:1].copy()), ptr(p4), None,
                                        ptr(out_h), None, None))
    out_d, _, _ = e4.refine_window(_t(tg), _t(sr), _t(dt), _t(ds), _t(host["K"][:1]), _t(p4), default_opts(n_iters=2), argmin=True)
    assert np.array_equal(out_h, out_d.cpu().numpy())


def test_intrinsics_changed_in_place_are_caught_on_the_device():
    """the host validates a device intrinsics buffer once per (pointer, count); when its CONTENTS later become non-pinhole the
    device-side guard poisons that call's results with NaN and the next call / synchronize reports TCSFM_E_INTRINSICS -- for the
    refinement (init_pair) and for the scale recovery (k_ground) alike"""
    from tightly_coupled_sfm_amd.engine import default_opts
    H, W, N = 24, 40, 2
    b = _pairs(N, H, W, seed0=1)
    e = _eng(H, W, N)
    d = _dev(b); p0 = _t(b["pose_init"])
    K = d[4].clone()
    good, _, _ = e.refine(d[0], d[1], d[2], d[3], K, p0, default_opts(n_iters=2))          # validates and caches this buffer
    Kgood = K.clone()
    K[:, 0, 1] = 0.5                                                                        # same address, skewed now
    bad, _, _ = e.refine(d[0], d[1], d[2], d[3], K, p0, default_opts(n_iters=2))           # asynchronous: no error yet
    with pytest.raises(RuntimeError, match="pinhole"):
        e.synchronize()
    assert torch.isnan(bad).all()
    K.copy_(Kgood)
    again, _, _ = e.refine(d[0], d[1], d[2], d[3], K, p0, default_opts(n_iters=2))         # the handle is healthy afterwards
    e.synchronize()
    assert torch.equal(again, good)
    # scale recovery: k_ground consumes K directly
    depth = d[2]
    s0 = e.scale_recovery(depth, K, 0.055)
    K[:, 1, 0] = 0.25
    s1 = e.scale_recovery(depth, K, 0.055)
    with pytest.raises(RuntimeError, match="pinhole"):
        e.synchronize()
    assert torch.isnan(s1).all() and torch.isfinite(s0).all()


def test_argument_errors_are_codes_not_crashes():
    """errors never cross the ABI as exceptions or faults: negative return code + tcsfm_last_error (SURVEY 8b 'Errors')"""
    import ctypes as C
    from tightly_coupled_sfm_amd.engine import default_opts
    H, W, N = 24, 40, 2
    b = _pairs(N, H, W, seed0=2)
    e = _eng(H, W, N)
    d = _dev(b); p0 = _t(b["pose_init"]); out = torch.empty_like(p0)
    P = lambda x: C.c_void_p(x.data_ptr())
    call = lambda o, n=N, tgt=d[0]: e.lib.tcsfm_refine(e._h, C.byref(o), n, P(tgt), P(d[1]), P(d[2]), P(d[3]), P(d[4]), P(p0), None, P(out), None, None)
    for bad, frag in ((dict(solver=5), "solver"), (dict(param=3), "param"), (dict(refine=9), "refine"), (dict(n_iters=-1), "n_iters"),
                      (dict(depth_is_disp=1, min_depth=0.0), "min_depth")):
        rc = call(default_opts(**bad))
        assert rc < 0 and frag in e.lib.tcsfm_last_error(e._h).decode(), (bad, rc)
    assert call(default_opts(), n=0) < 0 and call(default_opts(), n=N + 1) < 0
    assert e.lib.tcsfm_refine(e._h, C.byref(default_opts()), N, None, P(d[1]), P(d[2]), P(d[3]), P(d[4]), P(p0), None, P(out), None, None) < 0
    assert e.lib.tcsfm_refine(e._h, None, N, P(d[0]), P(d[1]), P(d[2]), P(d[3]), P(d[4]), P(p0), None, P(out), None, None) < 0
    dep = torch.empty_like(d[2])
    rc = e.lib.tcsfm_refine_dense(e._h, C.byref(default_opts(w_dc=0.1)), N, P(d[0]), P(d[1]), P(d[2]), P(d[3]), P(d[4]), P(p0), P(out), P(dep), None)
    assert rc < 0 and "w_dc" in e.lib.tcsfm_last_error(e._h).decode()
    rc = e.lib.tcsfm_refine_dense(e._h, C.byref(default_opts(param=1)), N, P(d[0]), P(d[1]), P(d[2]), P(d[3]), P(d[4]), P(p0), P(out), P(dep), None)
    assert rc < 0
    h = C.c_void_p()
    assert e.lib.tcsfm_create(C.byref(h), 0, 2, 2, 1) < 0 and b"sizes" in e.lib.tcsfm_last_error(None)
    # ... and the handle is still healthy afterwards
    assert call(default_opts()) == 0
    torch.cuda.synchronize()
    assert torch.isfinite(out).all()


def test_full_size_against_the_reference_run(oracle64):
    """BASELINE size (640x192): the HIP path against the REFERENCE's own float64 run on the same seeded inputs (golden
    full192x640: cost, mask count, autograd gradient, strided samples of the maps)"""
    from tightly_coupled_sfm_amd import synth
    g = load_golden("full192x640")
    H, W = 192, 640
    p = synth.make_pair(H, W, seed=0)
    chk = np.array([p[k].astype(np.float64).sum() for k in ("tgt", "src", "depth_t", "depth_s")])
    assert np.allclose(chk, g["in_checksum"], rtol=0, atol=1e-6), "synthetic generator drifted from the fixture"
    e = _eng(H, W, 1)
    args = (_t(p["tgt"][None]), _t(p["src"][None]), _t(p["depth_t"][None, None]), _t(p["depth_s"][None, None]))
    pose, K = _t(g["pose"][None]), _t(p["K"][None])
    lin = e.linearize(*args, K, pose)
    assert abs(lin["cost"][0] - float(g["f64_cost"])) < 2e-6 * float(g["f64_cost"])
    assert abs(lin["n_mask"][0] - float(g["f64_n_mask"])) <= 3                       # fp32 ties on the auto-mask threshold
    gp = oracle64.euler_left_jacobian(g["pose"]).T @ lin["g"][0]                     # d/d xi -> d/d pose (chain rule only)
    assert _maxabs(gp, g["f64_grad_pose"]) < 2e-4 * np.abs(g["f64_grad_pose"]).max()  # == reference autograd
    r = e.compute_photometric_error(args[0], args[1], args[2], args[3], pose, K)
    diff = r["diff_img"][0, 0].cpu().numpy(); wt = r["weight_mask"][0, 0].cpu().numpy()
    assert _maxabs(diff[::7, ::7], g["f64_diff_sub"]) < 2e-5 and _maxabs(wt[::7, ::7], g["f64_weight_sub"]) < 2e-5   # fp32 SSIM
    assert _maxabs(r["img_rec"][0].cpu().numpy()[:, ::7, ::7], g["f64_rec_sub"]) < 2e-6
    m = r["valid_mask"][0, 0].cpu().numpy()
    assert np.mean(m[::7, ::7] != g["f64_mask_sub"]) < 1e-3
    assert abs(float(diff.astype(np.float64).sum()) - float(g["f64_sum_diff"])) < 1e-5 * float(g["f64_sum_diff"])


def test_plain_c_caller(tmp_path):
    """examples/c_caller.c: a C99 program using only include/tcsfm.h with host arrays (no torch in the process)"""
    import subprocess
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from test_abi_cpu import _build_c_caller
    exe = _build_c_caller(tmp_path)
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    lines = [l for l in r.stdout.splitlines() if l.startswith("pair")]
    assert len(lines) == 2
    for l in lines:
        c0, c1 = (float(x) for x in l.split("cost")[1].split("pose")[0].replace("->", " ").split())
        assert c1 < 0.5 * c0, l                                   # the refinement found the parallax
    tx = [float(l.split("[")[1].split(",")[0]) for l in lines]
    assert tx[0] * tx[1] < 0 and min(abs(tx[0]), abs(tx[1])) > 3e-3          # opposite shifts -> opposite translations


def test_generate_loss_surface_drop_in_vs_reference_golden():
    """the plot_loss_surface.generate_loss_surface mirror: same arguments and result dict as the reference (golden G7)"""
    from tightly_coupled_sfm_amd.plot_loss_surface import generate_loss_surface
    g = load_golden("sweep48x160")
    data = (_t(g["tgt"][None]), [_t(g["src"][None])], None, None, None, _t(g["K"][None]), None, None, None, None, None)
    depths = [_t(g["depth_t"][None, None]), _t(g["depth_s"][None, None])]
    r = generate_loss_surface(data, depths, _t(g["pose"][None]).clone(), sample_trans=True, sample_yaw=True)
    assert np.array_equal(r["delta_list"], g["delta_list"]) and np.array_equal(r["delta_list_yaw"], g["delta_list_yaw"])
    assert abs(r["original_error"] - float(g["original_error"])) < 2e-5 * float(g["original_error"])
    for mine, ref in ((r["reconstruction_errors"], g["errors"]), (r["reconstruction_errors_yaw"], g["errors_yaw"])):
        rel = np.abs(mine - ref) / ref
        assert mine.shape == ref.shape and np.median(rel) < 2e-5 and rel.max() < 2e-3
    step_t, step_y = g["delta_list"][1] - g["delta_list"][0], g["delta_list_yaw"][1] - g["delta_list_yaw"][0]
    assert abs(float(r["best_trans_delta"]) - float(g["best_trans_delta"])) <= 1.01 * step_t
    assert abs(float(r["best_yaw_delta"]) - float(g["best_yaw_delta"])) <= 1.01 * step_y
    assert r["best_pose_vec"].shape == (1, 6) and r["best_error"] <= r["original_error"]


def test_reference_named_drop_ins_vs_goldens():
    """module-level functions with the reference's names / arguments / return structure:
    train_mono.solve_pose_iteratively (G4), helpers.compute_photometric_error (G3), stn.inverse_warp2 (G1),
    losses.SSIM_Loss (G2), learning_helpers.disp_to_depth (G8)"""
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from tightly_coupled_sfm_amd import helpers, learning_helpers, losses, stn, train_mono
    g = load_golden("batch24x40")
    S, B = g["sources"].shape[:2]

    class ConstPose(torch.nn.Module):          # the stand-in the golden was generated with (tests/golden/make_golden.py)
        def __init__(self, first, corr):
            super().__init__(); self.first, self.corr, self.calls = first, corr, 0
        def forward(self, x):
            self.calls += 1
            return self.first.clone() if self.calls == 1 else self.corr.clone()

    depths = [_t(g["depths"][i]) for i in range(S + 1)]
    for iters in (1, 4):
        pm = ConstPose(_t(g["first"]), _t(g["corr"]))
        poses, poses_inv, out = train_mono.solve_pose_iteratively(iters, depths, pm, _t(g["target"]), [_t(s) for s in g["sources"]], _t(g["K"]),
                                                                  return_errors=True)
        assert pm.calls == iters and len(poses) == S and tuple(poses[0].shape) == (B, 6)
        assert _maxabs(torch.stack(poses).cpu().numpy(), g[f"it{iters}_poses"]) < 1e-6
        assert _maxabs(torch.stack(poses_inv).cpu().numpy(), g[f"it{iters}_poses_inv"]) < 1e-6
        for d in ("fwd", "inv"):
            o = out[d]
            assert set(o) == {"diff_img", "img_rec", "valid_mask", "weight_mask", "poses", "auto_mask_error", "auto_mask"}
            assert _maxabs(o["diff_img"].cpu().numpy(), g[f"it{iters}_{d}_diff_img"]) < 3e-5
            assert _maxabs(o["weight_mask"].cpu().numpy(), g[f"it{iters}_{d}_weight_mask"]) < 1e-4
            assert _maxabs(o["img_rec"].cpu().numpy(), g[f"it{iters}_{d}_img_rec"]) < 1e-4
            assert _maxabs(o["poses"].cpu().numpy(), g[f"it{iters}_{d}_poses"]) < 1e-6
            assert (o["valid_mask"].cpu().numpy() != g[f"it{iters}_{d}_valid_mask"]).mean() <= 0.003
            assert (o["auto_mask"].cpu().numpy() != g[f"it{iters}_{d}_auto_mask"]).mean() <= 0.003
    s = load_golden("s24x40")
    H, W = s["tgt"].shape[1:]
    k = 0
    tgt, src, dt, ds, K = _t(s["tgt"][None]), _t(s["src"][None]), _t(s["depth_t"][None, None]), _t(s["depth_s"][None, None]), _t(s["K"][None])
    pose = _t(s["poses"][k][None])
    r = helpers.compute_photometric_error(tgt, src, dt, ds, pose, K)
    assert set(r) == {"diff_img", "img_rec", "valid_mask", "weight_mask", "poses"}
    assert _maxabs(r["diff_img"][0, 0].cpu().numpy(), s["f64_diff"][k]) < 3e-5 and _maxabs(r["weight_mask"][0, 0].cpu().numpy(), s["f64_weight"][k]) < 1e-4
    rec, valid, pd, cd = stn.inverse_warp2(src, dt, ds, -pose, K, "zeros")
    assert _maxabs(rec[0].cpu().numpy(), s["f64_rec"][k]) < 1e-4 and _maxabs(cd[0, 0].cpu().numpy(), s["f64_comp_depth"][k]) < 1e-5
    assert _maxabs(losses.SSIM_Loss()(tgt, src)[0].cpu().numpy(), s["f64_ssim_ts"]) < 3e-5
    assert _maxabs(stn.pose_vec2mat(-pose)[0].cpu().numpy().astype(np.float64), np.asarray(_pose_T(s["poses"][k]))) < 1e-6
    from tightly_coupled_sfm_amd import dnet_layers
    sc = load_golden("scale48x160")
    K4 = np.tile(np.eye(4, dtype=np.float32), (2, 1, 1)); K4[:, :3, :3] = sc["K"]
    scale = dnet_layers.ScaleRecovery(2, 48, 160).to("cuda")(_t(sc["depth"][:, None]), _t(K4), float(sc["cam_height"]))
    assert abs(float(scale) - float(sc["f32_scale"][0])) < 2e-5 * float(sc["f32_scale"][0])
    hp = load_golden("helpers")
    sd, dep = learning_helpers.disp_to_depth(_t(hp["disp"]), 0.06, 2.67)
    assert _maxabs(sd.cpu().numpy() / hp["scaled_disp"], 1.0) < 1e-6 and _maxabs(dep.cpu().numpy() / hp["depth"], 1.0) < 1e-6


def _pose_T(pose):
    from tightly_coupled_sfm_amd import synth
    return synth.pose_to_T(pose)


def test_dense_window_mode_vs_oracle(oracle64):
    """dense mode in window form (B targets x S sources, every directed pair refines its own copy of its target's depth), with
    and without the min over the sources, against the float64 oracle"""
    from oracle.oracle import default_opts as oopts
    from tightly_coupled_sfm_amd.engine import default_opts
    B, S, H, W = 1, 2, 96, 320
    w = _window(B, S, H, W)
    e = _eng(H, W, 2 * S * B)
    tg, sr, dt, ds, K, p0 = (_t(w[k]) for k in ("target", "sources", "depth_t", "depth_s", "K", "first"))
    o = default_opts(n_iters=3, w_dc=0.0, min_depth=0.06, max_depth=2.67, dense_joint=0)       # the per-pair-copy mode (joint: test_gpu_joint_dense.py)
    # without the selection it is the pair form on the stacked tensors
    pw, dw, _ = e.refine_dense_window(tg, sr, dt, ds, K, p0, o, argmin=False)
    T = tg.repeat(S, 1, 1, 1); Sx = sr.reshape(S * B, 3, H, W); Dt = dt.repeat(S, 1, 1, 1); Ds = ds.reshape(S * B, 1, H, W)
    pp, dp, _ = e.refine_dense(torch.cat([T, Sx]), torch.cat([Sx, T]), torch.cat([Dt, Ds]), torch.cat([Ds, Dt]), K.repeat(2 * S, 1, 1), p0, o)
    assert torch.equal(pw, pp) and torch.equal(dw, dp)
    # with it: parity with the oracle (pose 1e-4, per-pixel depth 1e-4, decisions replayed and bounded)
    r = PU.replay_window(e, oracle64, w, o, oopts(n_iters=3), _t, argmin=True, dense=True)
    st = r["stats"]
    assert np.all(st[:S * B, 0, 2] < 0.8 * H * W) and not np.array_equal(r["pose"][:S * B].astype(np.float32), pw.cpu().numpy()[:S * B])


@pytest.mark.parametrize("H,W", [(17, 33), (16, 32), (31, 65), (50, 70), (5, 9), (33, 16)])
def test_ragged_and_tiny_frames_vs_oracle(H, W, oracle64):
    """tile-boundary cases: frames smaller than a tile, one pixel over a tile edge, odd sizes, portrait"""
    from oracle.oracle import default_opts as oopts
    from tightly_coupled_sfm_amd.engine import default_opts
    N = 2
    b = _pairs(N, H, W, seed0=70 + H)
    e = _eng(H, W, N)
    d = _dev(b); p0 = _t(b["pose_init"])
    lin = e.linearize(*d, p0)
    pose, _, st = e.refine(*d, p0, default_opts(n_iters=2), stats=True)
    r = e.compute_photometric_error(d[0], d[1], d[2], d[3], p0, d[4])
    for n in range(N):
        args = (b["tgt"][n], b["src"][n], b["depth_t"][n, 0], b["depth_s"][n, 0])
        ref = oracle64.linearize(*args, b["pose_init"][n], b["K"][n], oopts())
        o = oracle64.photometric(*args, b["pose_init"][n], b["K"][n])
        assert _maxabs(r["diff_img"][n, 0].cpu().numpy(), o["diff"]) < 3e-5 and _maxabs(r["img_rec"][n].cpu().numpy(), o["rec"]) < 1e-5
        if ref["n_mask"] < 8:
            continue                                          # nothing to fit on a handful of pixels
        assert abs(lin["n_mask"][n] - ref["n_mask"]) <= 2 and abs(lin["cost"][n] - ref["cost"]) < 1e-3 * ref["cost"] + 1e-7
        if lin["n_mask"][n] == ref["n_mask"]:
            assert _maxabs(lin["g"][n], ref["g"]) < 5e-4 * np.abs(ref["g"]).max() and _maxabs(lin["H"][n], ref["H"]) < 5e-4 * np.abs(ref["H"]).max()
    assert torch.isfinite(pose).all()


def test_refine_window_three_sources_vs_oracle(oracle64):
    """S = 3: the selection keeps the FIRST minimum over the sources (two of the three sources see the same motion here, so
    near-ties between them are common) -- in-kernel selection against the oracle"""
    from oracle.oracle import default_opts as oopts
    from tightly_coupled_sfm_amd.engine import default_opts
    B, S, H, W = 1, 3, 64, 208
    w = _window(B, S, H, W)
    e = _eng(H, W, 2 * S * B)
    r = PU.replay_window(e, oracle64, w, default_opts(n_iters=3), oopts(n_iters=3), _t, argmin=True)
    st = r["stats"]
    assert np.all(st[:S * B, 0, 2] > 0) and st[:S * B, 0, 2].sum() < 0.98 * H * W
    _, _, rst = oracle64.refine_window(w["target"], w["sources"], w["depth_t"][:, 0], w["depth_s"][:, :, 0], w["K"], w["first"],
                                       oopts(n_iters=3), argmin=True)
    assert np.max(np.abs(st[:, :3, 2] - rst[:, :3, 2])) <= 8                 # a few fp32-vs-f64 tie decisions at most


@pytest.mark.parametrize("window", [False, True], ids=["pairs", "window+argmin"])
def test_dense_lm_vs_oracle(oracle64, window):
    """Levenberg-Marquardt in dense mode: accept / reject on the cost with the pose AND the depth map rolled back on a reject,
    final cost check -- against the float64 oracle (pair form, and window form with the min over sources)"""
    from oracle.oracle import default_opts as oopts
    from tightly_coupled_sfm_amd.engine import default_opts
    H, W, iters = 96, 320, 5
    o = default_opts(n_iters=iters, solver=1, lambda0=1e-3, w_dc=0.0, min_depth=0.06, max_depth=2.67)
    oo = oopts(n_iters=iters, solver=1, lambda0=1e-3)
    if window:
        B, S = 1, 2
        w = _window(B, S, H, W)
        w["depth_t"] = (w["depth_t"] * (1 + 0.02 * np.sin(np.arange(W) / 11.0))[None, None, None, :]).astype(np.float32)
        r = PU.replay_window(_eng(H, W, 2 * S * B), oracle64, w, o, oo, _t, argmin=True, dense=True)
    else:
        N = 2
        b = _pairs(N, H, W, seed0=31, both=True)
        d0 = (b["depth_t"] * (1 + 0.03 * np.sin(np.arange(W) / 9.0))[None, None, None, :]).astype(np.float32)
        far = np.stack([__import__("tightly_coupled_sfm_amd").synth.perturb_pose(g, 5 + i, sigma_t=0.002, sigma_r=0.0006) for i, g in enumerate(b["pose_gt"])]).astype(np.float32)
        r = PU.replay_dense_pairs(_eng(H, W, N), oracle64, b, d0, o, oo, _t, poses=far)
    assert np.all(r["stats"][:, -1, 0] <= r["stats"][:, 0, 0])                                   # LM never ends above where it started


def test_dense_lm_rejects_roll_back_pose_and_depth(oracle64):
    """cases where Levenberg-Marquardt REJECTS trials inside the loop and at the final check (found by scanning seeds with the
    oracle): the pose and the depth map must come back from the accepted state exactly as in the float64 oracle"""
    from oracle.oracle import default_opts as oopts
    from tightly_coupled_sfm_amd import synth
    from tightly_coupled_sfm_amd.engine import default_opts
    H, W, iters, seeds = 48, 160, 8, (21, 29, 31, 20)
    ps = [synth.make_pair(H, W, seed=s) for s in seeds]
    init = np.stack([synth.perturb_pose(p["pose_gt"], s, sigma_t=0.004, sigma_r=0.0012) for p, s in zip(ps, seeds)]).astype(np.float32)
    d0 = np.stack([(p["depth_t"] * (1 + 0.05 * np.sin(np.arange(W) / 7.0))[None, :]).astype(np.float32) for p in ps])[:, None]
    b = {k: np.stack([p[k] for p in ps]) for k in ("tgt", "src", "K")}
    b["depth_s"] = np.stack([p["depth_s"] for p in ps])[:, None]
    kw = dict(n_iters=iters, solver=1, lambda0=1e-4, lambda_min=1e-6)
    r = PU.replay_dense_pairs(_eng(H, W, len(seeds)), oracle64, b, d0, default_opts(w_dc=0.0, min_depth=0.06, max_depth=2.67, **kw),
                              oopts(**kw), _t, poses=init)
    dec = r["decide"]
    assert int((dec[1:iters] == 0).any(0).sum()) >= 2 and int((dec[iters] == 0).sum()) >= 1   # rejects inside the loop AND at the final check
    # the free-running oracle takes the same decisions in at least two of the rejecting cases (they are decisive, not ties)
    matched = 0
    for n, p in enumerate(ps):
        _, _, rst = oracle64.refine_dense(p["tgt"], p["src"], d0[n, 0], p["depth_s"], init[n], p["K"], oopts(**kw), lambda_depth=1.0, w_prior=10.0)
        matched += int(np.allclose(r["stats"][n, :, 3], rst[:, 3], rtol=1e-5) and (dec[:, n] == 0).any())
    assert matched >= 2


def test_fuzz_options_vs_oracle(oracle64):
    """seeded random sweep over sizes and option combinations (solver, chart, refine mode, depth consistency, auto-mask,
    iteration count, damping): every configuration against the float64 oracle at 1e-4 (decisions replayed and checked)"""
    from oracle.oracle import default_opts as oopts
    from tightly_coupled_sfm_amd.engine import default_opts
    rng = np.random.default_rng(2024)
    flips = lm = 0
    for case in range(16):
        H, W = int(rng.integers(20, 72)), int(rng.integers(36, 150))
        kw = dict(n_iters=int(rng.integers(1, 6)), solver=int(rng.integers(0, 2)), param=int(rng.integers(0, 2)),
                  w_dc=float(rng.choice([0.0, 0.15])), automask=int(rng.integers(0, 2)), lambda0=float(rng.choice([1e-4, 1e-3, 1e-2])))
        refine = int(rng.integers(0, 2))
        N = 2
        b = _pairs(N, H, W, seed0=300 + case, both=bool(rng.integers(0, 2)))
        ls0 = rng.normal(scale=0.03, size=N).astype(np.float32)
        r = PU.replay_pairs(_eng(H, W, N), oracle64, b, default_opts(refine=refine, **kw), oopts(nparam=6 + refine, **kw), _t, log_scale=ls0)
        flips += r["mask_flips"]; lm += r["lm_flips"]
    assert flips <= 16 and lm <= 4           # decisions differ from the float64 oracle's own only on a handful of ties


def test_fuzz_window_and_dense_vs_oracle(oracle64):
    """seeded random sweep over window shapes (B, S), selection on/off, solver, pose / dense mode at small sizes"""
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import standins
    from oracle.oracle import default_opts as oopts
    from tightly_coupled_sfm_amd.engine import default_opts
    rng = np.random.default_rng(77)
    flips = lm = 0
    for case in range(8):
        B, S = int(rng.integers(1, 3)), int(rng.integers(1, 4))
        H, W = int(rng.integers(24, 56)), int(rng.integers(48, 120))
        dense, argmin = bool(rng.integers(0, 2)), bool(rng.integers(0, 2))
        kw = dict(n_iters=int(rng.integers(1, 5)), solver=int(rng.integers(0, 2)), lambda0=float(rng.choice([1e-4, 1e-3])))
        w = standins.make_window(B, S, H, W, seed0=400 + case)
        w["depth_t"] = oracle64.disp_to_depth(w["disp_t"], 0.06, 2.67)[1].astype(np.float32)
        w["depth_s"] = oracle64.disp_to_depth(w["disp_s"], 0.06, 2.67)[1].astype(np.float32)
        o = default_opts(w_dc=0.0, min_depth=0.06, max_depth=2.67, **kw) if dense else default_opts(**kw)
        r = PU.replay_window(_eng(H, W, 2 * S * B), oracle64, w, o, oopts(**kw), _t, argmin=argmin, dense=dense)
        flips += r["mask_flips"]; lm += r["lm_flips"]
    assert flips <= 40 and lm <= 4


# ------------------------------------------------------------------------------------------------------------------
# BASELINE.json configs at their full size (640x192), end to end, against the float64 oracle
def test_config2_bench_workload_full_size(oracle64):
    """BASELINE config 2 exactly as bench.py runs it: one window of B=1 target / S=1 source = the fwd + inv directed pairs,
    640x192, 4 Gauss-Newton iterations of the 6-DoF pose -- free-running oracle and decision replay, both at 1e-4"""
    from oracle.oracle import default_opts as oopts
    from tightly_coupled_sfm_amd.engine import default_opts
    H, W = 192, 640
    b = _pairs(2, H, W, seed0=0, both=True)               # bench.py: synth.make_batch(2, H, W, seed0=100*rank, both_directions=True)
    e = _eng(H, W, 2)
    r = PU.replay_pairs(e, oracle64, b, default_opts(n_iters=4), oopts(n_iters=4), _t)
    for n in range(2):
        rp, _, rst = oracle64.refine(b["tgt"][n], b["src"][n], b["depth_t"][n, 0], b["depth_s"][n, 0], b["pose_init"][n], b["K"][n], oopts(n_iters=4))
        PU.assert_pose(r["pose"][n], rp, ("free", n))
        assert np.max(np.abs(r["stats"][n, :4, 2] - rst[:4, 2])) <= 3        # mask counts: a few tie pixels of 122 880
        assert np.max(np.abs(r["stats"][n, :4, 0] - rst[:4, 0]) / rst[:4, 0]) < 2e-5
    assert np.all(r["stats"][:, 3, 0] < 0.5 * r["stats"][:, 0, 0])           # and the refinement does its job


def test_config1_demo_minibatch_shape(oracle64):
    """BASELINE config 1's shape (run_sample_optimization_demo.py:87-88): minibatch B=3, S=2 sources -> 12 directed pairs at 640x192,
    ONE iteration, the reference's default loss options (min over sources, auto-mask, depth consistency 0.15)"""
    from oracle.oracle import default_opts as oopts
    from tightly_coupled_sfm_amd.engine import default_opts
    B, S, H, W = 3, 2, 192, 640
    w = _window(B, S, H, W)
    e = _eng(H, W, 2 * S * B)
    r = PU.replay_window(e, oracle64, w, default_opts(n_iters=1, w_dc=0.15), oopts(n_iters=1, w_dc=0.15), _t, argmin=True)
    rp, _, rst = oracle64.refine_window(w["target"], w["sources"], w["depth_t"][:, 0], w["depth_s"][:, :, 0], w["K"], w["first"],
                                        oopts(n_iters=1, w_dc=0.15), argmin=True)
    # one iteration from identical poses: the free-running oracle agrees at 1e-4 too wherever it took the SAME decisions; a pair in
    # which a tie pixel was decided differently is a (slightly) different problem -- the replay above vouches for its arithmetic at
    # 1e-4, and parity_util has already asserted that every such pixel is a near-tie and that they are few
    own = oracle64.window_select(w["target"], w["sources"], w["depth_t"][:, 0], w["depth_s"][:, :, 0], w["K"], w["first"][:S * B],
                                 oopts(n_iters=1, w_dc=0.15)) > 0.5
    same = [np.array_equal((r["bits"][0, n] & 1) > 0, own[n]) for n in range(S * B)] + \
           [r["stats"][n, 0, 2] == rst[n, 0, 2] for n in range(S * B, 2 * S * B)]
    assert sum(same) >= S * B                              # (most pairs: ties are rare)
    for n in range(2 * S * B):
        if same[n]:
            PU.assert_pose(r["pose"][n], rp[n], ("free", n))
    assert np.max(np.abs(r["stats"][:, 0, 2] - rst[:, 0, 2])) <= 6
    assert np.all(r["stats"][:S * B, 0, 2] > 0.02 * H * W)                     # both sources win somewhere


def test_config3_per_gpu_shard_8_windows(oracle64):
    """BASELINE config 3's per-GPU shard: 8 windows (B=8, S=1 -> 16 directed pairs) at 640x192 in ONE window-form call, 4 GN
    iterations: equal to the pair form bit for bit, equal to 8 separate B=1 calls bit for bit (what the other ranks compute),
    and within 1e-4 of the float64 oracle"""
    from oracle.oracle import default_opts as oopts
    from tightly_coupled_sfm_amd.engine import default_opts
    B, S, H, W = 8, 1, 192, 640
    w = _window(B, S, H, W)
    e = _eng(H, W, 2 * S * B)
    o = default_opts(n_iters=4)
    r = PU.replay_window(e, oracle64, w, o, oopts(n_iters=4), _t, argmin=True)
    tg, sr, dt, ds, K, p0 = (_t(w[k]) for k in ("target", "sources", "depth_t", "depth_s", "K", "first"))
    pw, _, _ = e.refine_window(tg, sr, dt, ds, K, p0, o)
    assert np.array_equal(pw.cpu().numpy(), r["pose"].astype(np.float32))
    e1 = _eng(H, W, 2)
    for bb in range(B):
        sel = torch.tensor([bb, B + bb], device=p0.device)
        p1, _, _ = e1.refine_window(tg[bb:bb + 1], sr[:, bb:bb + 1], dt[bb:bb + 1], ds[:, bb:bb + 1], K[bb:bb + 1], p0[sel].contiguous(), o)
        assert torch.equal(p1, pw[sel])                                        # batch independence: a shard is the sum of its windows


def test_handles_release_their_memory():
    """create / use / destroy many handles (all lazily allocated scratch exercised): device memory comes back"""
    import gc
    from tightly_coupled_sfm_amd.engine import Engine, default_opts
    H, W, N = 96, 320, 4
    w = _window(1, 2, H, W)
    args = tuple(_t(w[k]) for k in ("target", "sources", "depth_t", "depth_s", "K", "first"))
    torch.cuda.synchronize()
    free0, _ = torch.cuda.mem_get_info()
    for i in range(40):
        e = Engine(H, W, N)
        e.refine_window(*args, default_opts(n_iters=1), argmin=True)
        e.refine_dense_window(*args, default_opts(n_iters=1, solver=1, w_dc=0.0, min_depth=0.06, max_depth=2.67), argmin=True)
        e.scale_recovery(args[2], args[4], 0.055)
        torch.cuda.synchronize()
        e.close()
    gc.collect(); torch.cuda.synchronize()
    free1, _ = torch.cuda.mem_get_info()
    assert free0 - free1 < 64 << 20, (free0, free1)          # 40 leaked handles would hold > 1 GB


def test_degenerate_inputs_stay_contained():
    """Garbage in one pair (NaN texels, non-positive depths, a NaN start pose, textureless images) never crashes or hangs, never
    leaks into the other pairs of the call (their results are bit-identical to a clean call), and a textureless pair -- singular
    normal equations -- comes back with its start pose"""
    from tightly_coupled_sfm_amd.engine import default_opts
    H, W = 48, 160
    b = _pairs(3, H, W, seed0=21)
    e = _eng(H, W, 3)
    o = default_opts(n_iters=3)
    clean = e.refine(*_dev(b), _t(b["pose_init"]), o)[0].cpu().numpy()

    def run(mut):
        bb = {k: v.copy() for k, v in b.items()}
        mut(bb)
        return e.refine(*_dev(bb), _t(bb["pose_init"]), o)[0].cpu().numpy()

    def nan_texels(bb): bb["src"][1, :, 10:20, 30:60] = np.nan
    def bad_depth(bb): bb["depth_t"][1, 0, ::3, ::5] = 0.0; bb["depth_s"][1, 0, 5:9] = -1.0
    def nan_pose(bb): bb["pose_init"][1, 2] = np.nan
    def flat(bb): bb["tgt"][1] = 0.5; bb["src"][1] = 0.5
    for mut in (nan_texels, bad_depth, nan_pose, flat):
        got = run(mut)
        assert np.array_equal(got[0], clean[0]) and np.array_equal(got[2], clean[2]), mut.__name__     # the neighbours are untouched
        if mut is flat:
            assert np.allclose(got[1], b["pose_init"][1], atol=1e-6), got[1]                             # no information -> no step
        if mut is nan_pose:
            assert np.isnan(got[1][:3]).all()                                                             # a NaN start pose is not laundered into numbers
    again = e.refine(*_dev(b), _t(b["pose_init"]), o)[0].cpu().numpy()
    assert np.array_equal(again, clean)                                                                   # and the handle is none the worse


def test_adjoint_form_of_the_ssim_gradient_is_the_same_gradient(tmp_path):
    """k_linearize<ADJ> (TCSFM_ADJOINT=1, kernels.h): the adjoint form of the 3x3-coupled SSIM gradient sums the same products in another
    order -- gradient and refined poses equal the default form's to rounding (measured slower, kept as an option: profiles/r04_adjoint_ab.txt)"""
    import subprocess, sys
    code = (
        "import numpy as np, torch, sys\n"
        "from tightly_coupled_sfm_amd import synth\n"
        "from tightly_coupled_sfm_amd.engine import Engine, default_opts\n"
        "b = synth.make_batch(4, 96, 320, seed0=7, both_directions=True)\n"
        "d = {k: torch.as_tensor(v).cuda() for k, v in b.items()}\n"
        "e = Engine(96, 320, 4)\n"
        "out = {}\n"
        "for tag, o in (('pose', default_opts(n_iters=4, w_dc=0.15)), ('scale', default_opts(n_iters=4, refine=1))):\n"
        "    lin = e.linearize(d['tgt'], d['src'], d['depth_t'], d['depth_s'], d['K'], d['pose_init'], o)\n"
        "    pose, ls, _ = e.refine(d['tgt'], d['src'], d['depth_t'], d['depth_s'], d['K'], d['pose_init'], o)\n"
        "    out[tag + '_g'] = lin['g']; out[tag + '_H'] = lin['H']; out[tag + '_pose'] = pose.cpu().numpy()\n"
        "np.savez(sys.argv[1], **out)\n")
    res = {}
    for adj in ("0", "1"):
        f = str(tmp_path / f"adj{adj}.npz")
        r = subprocess.run([sys.executable, "-c", code, f], env=dict(os.environ, TCSFM_ADJOINT=adj), capture_output=True, text=True, timeout=600, cwd=REPO)
        assert r.returncode == 0, r.stderr[-2000:]
        res[adj] = np.load(f)
    for k in res["0"].files:
        a, b = res["0"][k], res["1"][k]
        tol = 2e-5 if k.endswith("_g") else (1e-6 if k.endswith("_H") else 1e-5)
        assert np.abs(a - b).max() <= tol * np.abs(a).max(), (k, np.abs(a - b).max(), np.abs(a).max())
    assert not np.array_equal(res["0"]["pose_g"], res["1"]["pose_g"])       # (it really is another code path)
