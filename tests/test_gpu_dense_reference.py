"""Dense window mode on the REFERENCE's own loss (window_rule = TCSFM_WINDOW_REFERENCE): optimizer.py:47-90 with the
poses of all directed pairs and one inverse-depth map per target as unknowns.
  * tcsfm_linearize_dense_window reproduces the reference's loss and its autograd gradients w.r.t. every pose AND the shared target depth
    (golden G13 `full`, `fullinit`; the depth gradient includes what the inverse pairs see through their bilinear sample of the map);
  * the Gauss-Newton iterates follow the float64 oracle (orc_refine_dense_ref, itself pinned on the same goldens to 1e-10) within the
    north-star tolerance, with the engine's discrete decisions replayed, at 240x320, 256x448 (S = 1) and 192x640 (S = 2)."""
import numpy as np
import pytest
import torch

from conftest import load_golden
from oracle.oracle import Oracle, default_opts as oracle_opts
from parity_util import check_dense_ref_flips

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def orc():
    return Oracle("f64")


def _dev(a):
    return torch.as_tensor(np.ascontiguousarray(a, dtype=np.float32)).cuda()


@pytest.mark.parametrize("name", ["winloss24x40", "winloss48x160"])
def test_linearize_dense_window_vs_reference_autograd_G13(name, orc):
    from tightly_coupled_sfm_amd.engine import Engine, default_opts
    g = load_golden(name)
    S, B = g["sources"].shape[:2]
    H, W = g["target"].shape[-2:]
    SB = S * B
    mind, maxd = (float(x) for x in g["min_max_depth"])
    rd = 1.0 / mind - 1.0 / maxd
    e = Engine(H, W, 2 * SB)
    t = dict(tgt=_dev(g["target"]), srcs=_dev(g["sources"]), depth_t=_dev(g["depth_t"]), depth_s=_dev(g["depth_s"]), K=_dev(g["K"]), pose=_dev(g["first"]))
    for tag, argmin, w_init in (("full", True, 0.0), ("noargmin_full", False, 0.0), ("fullinit", True, 0.1), ("fullinit_smooth", True, 0.1), ("full_pc", True, 0.0)):
        w_smooth = 2.0 if tag == "fullinit_smooth" else 0.0          # l_smooth_weight x get_smooth_loss (optimizer.py:92-93)
        w_pc = 0.1 if tag == "full_pc" else 0.0                      # 0.1 (poses + poses_inv).abs().mean() (optimizer.py:95-96)
        o = default_opts(n_iters=1, w_dc=0.15, irls_eps=1e-7, prior_init=w_init, min_depth=mind, max_depth=maxd, w_smooth=w_smooth, w_pose_consist=w_pc)
        d0 = None if w_init == 0 else 1.0 / (1.0 / maxd + rd * g["sig_t0"])          # the golden's "initial" disparity as the prior's centre
        L = e.linearize_dense_window(t["tgt"], t["srcs"], t["depth_t"], t["depth_s"], t["K"], t["pose"], o, argmin=argmin,
                                     depth0=None if d0 is None else _dev(d0[:, None]))
        ref_loss = float(g[f"{tag}_loss"])
        assert abs(L["loss"] - ref_loss) < 1e-5 * ref_loss, (tag, L["loss"], ref_loss)
        gp = np.stack([orc.euler_left_jacobian(g["first"][m]).T @ L["g_pose"][m] for m in range(2 * SB)])
        ref_gp = g[f"{tag}_grad_pose"]
        assert np.abs(gp - ref_gp).max() < 2e-4 * np.abs(ref_gp).max(), (tag, np.abs(gp - ref_gp).max(), np.abs(ref_gp).max())
        g_rho = L["g_rho"][:, 0].cpu().numpy().astype(np.float64)
        if tag == "full":            # d / d depth = -rho^2 d / d rho
            gd, ref = -g_rho / g["depth_t"][:, 0] ** 2, g["full_grad_depth_t"]
            assert np.abs(gd - ref).max() < 2e-4 * np.abs(ref).max(), (tag, np.abs(gd - ref).max(), np.abs(ref).max())
        if tag in ("fullinit", "fullinit_smooth"):        # d / d sigma = r d / d rho: the whole loss incl. the SSIM prior (and l_smooth), through the shared depth
            gs, ref = g_rho * rd, g[f"{tag}_grad_sig_t"]
            assert np.abs(gs - ref).max() < 2e-4 * np.abs(ref).max(), (tag, np.abs(gs - ref).max(), np.abs(ref).max())
        # the engine against the oracle's restatement at the same point (float64, pinned to 1e-10 on the same goldens)
        oo = oracle_opts(n_iters=1, w_dc=0.15, irls_eps=1e-7, w_smooth=w_smooth, w_pose_consist=w_pc)
        if tag == "full_pc":
            assert L["pose_consist"] > 1e-6 and abs(L["loss"] - L["pose_consist"] - float(g["full_loss"])) < 1e-5 * ref_loss
        Lo = orc.linearize_dense_ref(g["target"], g["sources"], g["depth_t"][:, 0], g["depth_s"][:, :, 0], g["K"], g["first"], oo, argmin=argmin,
                                     w_init=w_init, depth0=d0, min_depth=mind, max_depth=maxd)
        assert abs(L["loss"] - Lo["loss"]) < 1e-5 * Lo["loss"] and L["K_f"] == Lo["K_f"] and L["K_i"] == Lo["K_i"]
        assert np.abs(L["g_pose"] - Lo["g_xi"]).max() < 2e-4 * np.abs(Lo["g_xi"]).max()
        assert np.abs(L["g_rho"][:, 0].cpu().numpy() - Lo["g_rho"]).max() < 2e-4 * np.abs(Lo["g_rho"]).max()


@pytest.mark.parametrize("name", ["winloss24x40", "winloss48x160"])
def test_gradient_wrt_the_source_depth_maps_vs_reference_autograd_G13(name, orc):
    """tcsfm_linearize_dense_window_sources: the gradient of the reference's loss w.r.t. the SOURCE depth maps -- leaves of the reference's
    optimize_depth_pred (optimizer.py:194-198) that the refinement holds fixed -- equals reference autograd (golden G13 `full_grad_depth_s`),
    with and without the min over the sources against the float64 oracle; the other outputs are the bits of tcsfm_linearize_dense_window"""
    from tightly_coupled_sfm_amd.engine import Engine, default_opts
    g = load_golden(name)
    S, B = g["sources"].shape[:2]
    H, W = g["target"].shape[-2:]
    mind, maxd = (float(x) for x in g["min_max_depth"])
    e = Engine(H, W, 2 * S * B)
    t = dict(tgt=_dev(g["target"]), srcs=_dev(g["sources"]), depth_t=_dev(g["depth_t"]), depth_s=_dev(g["depth_s"]), K=_dev(g["K"]), pose=_dev(g["first"]))
    for argmin in (True, False):
        o = default_opts(n_iters=1, w_dc=0.15, irls_eps=1e-7, prior_init=0.0, min_depth=mind, max_depth=maxd)
        L = e.linearize_dense_window(t["tgt"], t["srcs"], t["depth_t"], t["depth_s"], t["K"], t["pose"], o, argmin=argmin, sources=True)
        L0 = e.linearize_dense_window(t["tgt"], t["srcs"], t["depth_t"], t["depth_s"], t["K"], t["pose"], o, argmin=argmin)
        assert L["loss"] == L0["loss"] and np.array_equal(L["g_pose"], L0["g_pose"]) and torch.equal(L["g_rho"], L0["g_rho"])
        gs = L["g_rho_src"][:, :, 0].cpu().numpy().astype(np.float64)
        Lo = orc.linearize_dense_ref(g["target"], g["sources"], g["depth_t"][:, 0], g["depth_s"][:, :, 0], g["K"], g["first"],
                                     oracle_opts(n_iters=1, w_dc=0.15, irls_eps=1e-7), argmin=argmin, min_depth=mind, max_depth=maxd)
        assert np.abs(gs - Lo["g_rho_s"]).max() < 2e-4 * np.abs(Lo["g_rho_s"]).max(), (argmin, np.abs(gs - Lo["g_rho_s"]).max(), np.abs(Lo["g_rho_s"]).max())
        if argmin:           # d / d depth = -rho^2 d / d rho against reference autograd itself
            gd, ref = -gs / g["depth_s"][:, :, 0] ** 2, g["full_grad_depth_s"]
            assert np.abs(gd - ref).max() < 2e-4 * np.abs(ref).max(), (np.abs(gd - ref).max(), np.abs(ref).max())
    e.close()


def _window(B, S, H, W, seed, bias=1.02):
    from tightly_coupled_sfm_amd import synth
    # B targets x S sources: target b with sources from independent pairs that share its target image / depth
    tg, dt, sr, ds, K, p0 = [], [], [[] for _ in range(S)], [[] for _ in range(S)], [], [[] for _ in range(S)]
    for bb in range(B):
        for s in range(S):
            sign = 1.0 if s == 0 else -1.0
            base = np.array([0.003, -0.002, 0.033, 0.002, -0.004, 0.0015]) * sign
            p = synth.make_pair(H, W, seed=seed + 7 * bb, pose_gt=base, dtype=np.float64)
            if s == 0:
                tg.append(p["tgt"]); dt.append(p["depth_t"] * bias); K.append(p["K"])
            sr[s].append(p["src"]); ds[s].append(p["depth_s"]); p0[s].append(synth.perturb_pose(p["pose_gt"], seed + s))
    fwd = np.concatenate([np.stack(x) for x in p0])
    return dict(tgt=np.stack(tg), srcs=np.stack([np.stack(x) for x in sr]), depth_t=np.stack(dt), depth_s=np.stack([np.stack(x) for x in ds]),
                K=np.stack(K), pose=np.concatenate([fwd, -fwd]))


@pytest.mark.parametrize("H,W,S,mind,maxd", [(240, 320, 1, 0.03, 3.0), (256, 448, 1, 0.03, 3.0), (192, 640, 2, 0.06, 2.67)])
def test_dense_reference_iterates_follow_the_oracle(H, W, S, mind, maxd, orc):
    from tightly_coupled_sfm_amd.engine import Engine, default_opts
    from tightly_coupled_sfm_amd import _lib
    B, n_it = 1, 3
    w = _window(B, S, H, W, seed=31)
    N = 2 * S * B
    e = Engine(H, W, N)
    o = default_opts(n_iters=n_it, w_dc=0.15, prior_init=0.1, min_depth=mind, max_depth=maxd, window_rule=_lib.WINDOW_REFERENCE, lambda_depth=1.0)
    t = {k: _dev(v) for k, v in w.items()}
    dt4, ds5 = t["depth_t"][:, None].contiguous(), t["depth_s"][:, :, None].contiguous()
    e.trace_begin(n_it, N)
    pose, depth, st = e.refine_dense_window(t["tgt"], t["srcs"], dt4, ds5, t["K"], t["pose"], o, stats=True, argmin=True)
    bits, _ = e.trace_end()
    pose = pose.cpu().numpy().astype(np.float64); depth = depth.cpu().numpy().astype(np.float64)
    f32 = lambda a: np.asarray(a, dtype=np.float32).astype(np.float64)          # the oracle starts from the SAME float32 inputs
    oo = oracle_opts(n_iters=n_it, w_dc=0.15)
    orc.flip_stats_reset()
    po, do, so = orc.refine_dense_ref(f32(w["tgt"]), f32(w["srcs"]), f32(w["depth_t"]), f32(w["depth_s"]), f32(w["K"]), f32(w["pose"]), oo, argmin=True,
                                      w_init=0.1, lambda_depth=1.0, min_depth=mind, max_depth=maxd, bits=bits.reshape(n_it, N, H * W))
    nf, hard = orc.flip_stats(n_it)
    check_dense_ref_flips(nf, hard, N * H * W)
    for m in range(N):
        et = np.linalg.norm(pose[m, :3] - po[m, :3]) / np.linalg.norm(po[m, :3]); er = np.linalg.norm(pose[m, 3:] - po[m, 3:]) / np.linalg.norm(po[m, 3:])
        assert et < 1e-4 and er < 1e-4, (m, et, er)
    for s in range(S):       # the S forward slots carry the same refined map; every pixel within 1e-4
        assert np.abs(depth[s * B:(s + 1) * B, 0] / do - 1).max() < 1e-4, (s, np.abs(depth[s * B:(s + 1) * B, 0] / do - 1).max())
    assert np.array_equal(depth[S * B:, 0], f32(w["depth_s"]).reshape(S * B, H, W).astype(np.float32).astype(np.float64))       # source depths: not unknowns
    # the loss the engine reports (forward group + inverse pairs) is the oracle's, and it falls
    assert np.all(np.diff(so[:, 0]) < 0), so[:, 0]
    assert np.abs(depth[0, 0] / f32(w["depth_t"])[0] - 1).max() > 1e-3          # the map really moved


@pytest.mark.parametrize("B,S,H,W,quarter", [(6, 1, 24, 40, False), (6, 1, 24, 40, True), (4, 1, 48, 160, False), (6, 2, 48, 160, False)])
def test_dense_reference_minibatches_on_an_exactly_sized_handle(B, S, H, W, quarter, orc):
    """with ONE source per target a call has max_pairs / 2 targets -- the per-target scratch of the joint kernels (state,
    step, accepted depth) was sized for max_pairs / 4 and k_solve_joint<1> indexed past it.  The reference's own minibatch (6 windows,
    run_sequential_optimization.py:186) on a handle created with exactly 2 S B pairs: poses and every pixel of every map follow the oracle"""
    from tightly_coupled_sfm_amd.engine import Engine, default_opts
    from tightly_coupled_sfm_amd import _lib
    n_it = 2
    w = _window(B, S, H, W, seed=91)
    N = 2 * S * B
    e = Engine(H, W, N)
    assert e.max_pairs == N
    o = default_opts(n_iters=n_it, w_dc=0.15, prior_init=0.1, min_depth=0.06, max_depth=2.67, window_rule=_lib.WINDOW_REFERENCE, lambda_depth=1.0,
                     depth_param=_lib.DEPTH_QUARTER if quarter else _lib.DEPTH_FULL)
    t = {k: _dev(v) for k, v in w.items()}
    dt4, ds5 = t["depth_t"][:, None].contiguous(), t["depth_s"][:, :, None].contiguous()
    e.trace_begin(n_it, N)
    pose, depth, _ = e.refine_dense_window(t["tgt"], t["srcs"], dt4, ds5, t["K"], t["pose"], o, stats=True, argmin=True)
    bits, _ = e.trace_end()
    pose = pose.cpu().numpy().astype(np.float64); depth = depth.cpu().numpy().astype(np.float64)
    f32 = lambda a: np.asarray(a, dtype=np.float32).astype(np.float64)
    orc.flip_stats_reset()
    fn = orc.refine_dense_ref_q if quarter else orc.refine_dense_ref
    res = fn(f32(w["tgt"]), f32(w["srcs"]), f32(w["depth_t"]), f32(w["depth_s"]), f32(w["K"]), f32(w["pose"]), oracle_opts(n_iters=n_it, w_dc=0.15),
             argmin=True, w_init=0.1, lambda_depth=1.0, min_depth=0.06, max_depth=2.67, bits=bits.reshape(n_it, N, H * W))
    po, do = res[0], res[1]
    nf, hard = orc.flip_stats(n_it)
    assert hard.sum() == 0, (nf, hard)
    for m in range(N):
        et = np.linalg.norm(pose[m, :3] - po[m, :3]) / np.linalg.norm(po[m, :3]); er = np.linalg.norm(pose[m, 3:] - po[m, 3:]) / np.linalg.norm(po[m, 3:])
        assert et < 1e-4 and er < 1e-4, (m, et, er)
    for s in range(S):
        assert np.abs(depth[s * B:(s + 1) * B, 0] / do - 1).max() < 1e-4, (s, np.abs(depth[s * B:(s + 1) * B, 0] / do - 1).max())
    # a second call on the same handle (scratch reused) returns the same bits
    pose2, depth2, _ = e.refine_dense_window(t["tgt"], t["srcs"], dt4, ds5, t["K"], t["pose"], o, stats=True, argmin=True)
    assert np.array_equal(pose2.cpu().numpy().astype(np.float64), pose) and np.array_equal(depth2.cpu().numpy().astype(np.float64), depth)
    e.close()


@pytest.mark.parametrize("B,H,W,S,mind,maxd,argmin", [(1, 240, 320, 1, 0.03, 3.0, True), (1, 192, 640, 2, 0.06, 2.67, True), (1, 256, 448, 1, 0.03, 3.0, True), (1, 128, 416, 3, 0.06, 2.67, True),
                                                       (2, 48, 160, 2, 0.06, 2.67, True),
                                                       (1, 48, 160, 3, 0.06, 2.67, True), (2, 48, 160, 2, 0.06, 2.67, False)])
def test_free_source_depth_maps_follow_the_oracle(B, H, W, S, mind, maxd, argmin, orc):
    """opts.free_source_depths: the SOURCE depth maps are unknowns as well (the reference's optimize_depth_pred optimises the disparities of
    target and sources, optimizer.py:194-198) -- every inverse pair a group of its pose and the source map it back-projects (the joint kernel /
    solve / update on the inverse views, the adjoint of the forward pairs' samples in its gradient).  Poses, the target map and every pixel of
    every source map follow orc_refine_dense_ref_free to 1e-4 with the engine's discrete decisions replayed; the loss falls from linearisation
    to linearisation"""
    from tightly_coupled_sfm_amd.engine import Engine, default_opts
    from tightly_coupled_sfm_amd import _lib
    n_it = 3
    w = _window(B, S, H, W, seed=31)
    N = 2 * S * B
    e = Engine(H, W, N)
    o = default_opts(n_iters=n_it, w_dc=0.15, prior_init=0.1, min_depth=mind, max_depth=maxd, window_rule=_lib.WINDOW_REFERENCE, lambda_depth=1.0,
                     free_source_depths=1)
    t = {k: _dev(v) for k, v in w.items()}
    dt4, ds5 = t["depth_t"][:, None].contiguous(), t["depth_s"][:, :, None].contiguous()
    e.trace_begin(n_it, N)
    pose, depth, st = e.refine_dense_window(t["tgt"], t["srcs"], dt4, ds5, t["K"], t["pose"], o, stats=True, argmin=argmin)
    bits, _ = e.trace_end()
    pose = pose.cpu().numpy().astype(np.float64); depth = depth.cpu().numpy().astype(np.float64)
    f32 = lambda a: np.asarray(a, dtype=np.float32).astype(np.float64)
    oo = oracle_opts(n_iters=n_it, w_dc=0.15)
    orc.flip_stats_reset()
    po, do, dso, so = orc.refine_dense_ref_free(f32(w["tgt"]), f32(w["srcs"]), f32(w["depth_t"]), f32(w["depth_s"]), f32(w["K"]), f32(w["pose"]), oo, argmin=argmin,
                                                w_init=0.1, lambda_depth=1.0, min_depth=mind, max_depth=maxd, bits=bits.reshape(n_it, N, H * W))
    nf, hard = orc.flip_stats(n_it)
    check_dense_ref_flips(nf, hard, N * H * W)
    for m in range(N):
        et = np.linalg.norm(pose[m, :3] - po[m, :3]) / np.linalg.norm(po[m, :3]); er = np.linalg.norm(pose[m, 3:] - po[m, 3:]) / np.linalg.norm(po[m, 3:])
        assert et < 1e-4 and er < 1e-4, (m, et, er)
    for s in range(S):
        assert np.abs(depth[s * B:(s + 1) * B, 0] / do - 1).max() < 1e-4, (s, np.abs(depth[s * B:(s + 1) * B, 0] / do - 1).max())
    src_gpu = depth[S * B:, 0].reshape(S, B, H, W)
    # every pixel of every source map within 1e-4 (measured at 192 x 640, S = 2: worst 1.7e-6 / 6.6e-6 / 2.1e-5 after one / two / three
    # iterations, the 99.99 % quantile 5e-6).  Until k_dref_scatter_src took the sign of cd - pd from the same fp32 expression as the forward
    # group's kernel, ONE bilinear cell stood out at 1.3e-4: a forward sample with |cd - pd| / (cd + pd) = 2e-8, zero to the group's kernel and
    # negative to the scatter (scripts/diag/free_source_pixel.py, appendix R4)
    dev = np.abs(src_gpu / dso - 1)
    assert dev.max() < 1e-4, np.sort(dev.ravel())[-6:]
    assert np.quantile(dev, 0.9999) < 2e-5
    assert np.abs(src_gpu / f32(w["depth_s"]) - 1).max() > 1e-3                    # the source maps really moved
    assert np.all(np.diff(so[:, 0]) < 0), so[:, 0]
    # without the flag the inverse slots return the inputs; under the PAIR rule (the library's own dense modes) the flag is refused
    o.free_source_depths = 0
    _, d_fix, _ = e.refine_dense_window(t["tgt"], t["srcs"], dt4, ds5, t["K"], t["pose"], o, stats=True, argmin=argmin)
    assert torch.equal(d_fix[S * B:, 0].reshape(S, B, H, W), t["depth_s"])
    with pytest.raises(RuntimeError):
        e.refine_dense_window(t["tgt"], t["srcs"], dt4, ds5, t["K"], t["pose"], default_opts(n_iters=1, min_depth=mind, max_depth=maxd, free_source_depths=1), argmin=argmin)
    e.close()


@pytest.mark.parametrize("name", ["winloss24x40", "winloss48x160"])
def test_quarter_resolution_gradient_vs_reference_autograd_G13(name, orc):
    """the reference's PARAMETRISATION (optimizer.py:194-198, 235-239): at the x4-upsampled quarter-resolution maps the engine's loss and
    pose gradients are the reference's, and its depth gradient carried through the transposed upsampling equals reference autograd w.r.t.
    the quarter-resolution leaf (golden G13 `qinit`: F.interpolate and loss.backward() of the reference run)"""
    from tightly_coupled_sfm_amd.engine import Engine, default_opts
    g = load_golden(name)
    S, B = g["sources"].shape[:2]
    H, W = g["target"].shape[-2:]
    SB = S * B
    mind, maxd = (float(x) for x in g["min_max_depth"])
    rd = 1.0 / mind - 1.0 / maxd
    depth_of = lambda sig: 1.0 / (1.0 / maxd + rd * sig)
    e = Engine(H, W, 2 * SB)
    o = default_opts(n_iters=1, w_dc=0.15, irls_eps=1e-7, prior_init=0.1, min_depth=mind, max_depth=maxd)
    depth_s = np.stack([depth_of(g["q_up"][:, 1 + s]) for s in range(S)])
    L = e.linearize_dense_window(_dev(g["target"]), _dev(g["sources"]), _dev(depth_of(g["q_up"][:, 0])[:, None]), _dev(depth_s[:, :, None]), _dev(g["K"]),
                                 _dev(g["first"]), o, argmin=True, depth0=_dev(depth_of(g["sig_t0"])[:, None]))
    ref_loss = float(g["qinit_loss"])
    assert abs(L["loss"] - ref_loss) < 1e-5 * ref_loss, (L["loss"], ref_loss)
    gp = np.stack([orc.euler_left_jacobian(g["first"][m]).T @ L["g_pose"][m] for m in range(2 * SB)])
    assert np.abs(gp - g["qinit_grad_pose"]).max() < 2e-4 * np.abs(g["qinit_grad_pose"]).max()
    g_rho = L["g_rho"][:, 0].cpu().numpy().astype(np.float64)
    gq = np.stack([orc.up4_adjoint(g_rho[b] * rd) for b in range(B)])
    ref = g["qinit_grad_q"][:, 0]
    assert np.abs(gq - ref).max() < 2e-4 * np.abs(ref).max(), (np.abs(gq - ref).max(), np.abs(ref).max())


@pytest.mark.parametrize("H,W,S,mind,maxd", [(240, 320, 1, 0.03, 3.0), (192, 640, 2, 0.06, 2.67)])
def test_quarter_resolution_iterates_follow_the_oracle(H, W, S, mind, maxd, orc):
    """depth_param = TCSFM_DEPTH_QUARTER: Gauss-Newton on the quarter-resolution unknown (k_qres_*: projection of the input, x4 upsampling,
    cell records through the transposed upsampling, lumped cell curvature, per-cell Schur complement) follows orc_refine_dense_ref_q -- poses and
    every pixel of the returned (upsampled) map within 1e-4, the engine's discrete decisions replayed"""
    from tightly_coupled_sfm_amd.engine import Engine, default_opts
    from tightly_coupled_sfm_amd import _lib
    B, n_it = 1, 3
    w = _window(B, S, H, W, seed=31)
    N = 2 * S * B
    e = Engine(H, W, N)
    o = default_opts(n_iters=n_it, w_dc=0.15, prior_init=0.1, min_depth=mind, max_depth=maxd, window_rule=_lib.WINDOW_REFERENCE, lambda_depth=1.0,
                     depth_param=_lib.DEPTH_QUARTER)
    t = {k: _dev(v) for k, v in w.items()}
    dt4, ds5 = t["depth_t"][:, None].contiguous(), t["depth_s"][:, :, None].contiguous()
    e.trace_begin(n_it, N)
    pose, depth, st = e.refine_dense_window(t["tgt"], t["srcs"], dt4, ds5, t["K"], t["pose"], o, stats=True, argmin=True)
    bits, _ = e.trace_end()
    pose = pose.cpu().numpy().astype(np.float64); depth = depth.cpu().numpy().astype(np.float64)
    f32 = lambda a: np.asarray(a, dtype=np.float32).astype(np.float64)
    oo = oracle_opts(n_iters=n_it, w_dc=0.15)
    orc.flip_stats_reset()
    po, do, so, rq = orc.refine_dense_ref_q(f32(w["tgt"]), f32(w["srcs"]), f32(w["depth_t"]), f32(w["depth_s"]), f32(w["K"]), f32(w["pose"]), oo, argmin=True,
                                            w_init=0.1, lambda_depth=1.0, min_depth=mind, max_depth=maxd, bits=bits.reshape(n_it, N, H * W))
    nf, hard = orc.flip_stats(n_it)
    check_dense_ref_flips(nf, hard, N * H * W)
    for m in range(N):
        et = np.linalg.norm(pose[m, :3] - po[m, :3]) / np.linalg.norm(po[m, :3]); er = np.linalg.norm(pose[m, 3:] - po[m, 3:]) / np.linalg.norm(po[m, 3:])
        assert et < 1e-4 and er < 1e-4, (m, et, er)
    for s in range(S):
        assert np.abs(depth[s * B:(s + 1) * B, 0] / do - 1).max() < 1e-4, (s, np.abs(depth[s * B:(s + 1) * B, 0] / do - 1).max())
    assert np.all(np.diff(so[:, 0]) < 0), so[:, 0]
    # the returned map is a x4 upsampling: its quarter-resolution projection upsampled again reproduces it (a full-resolution map would not)
    back = 1.0 / orc.up4(rq[0])
    assert np.abs(depth[0, 0] / back - 1).max() < 1e-4 and np.abs(rq[0] - orc.down4(1.0 / f32(w["depth_t"])[0])).max() > 1e-4
    # full resolution on the same inputs is a different iterate; n_iters = 0 returns the projected-and-upsampled input
    o0 = default_opts(n_iters=0, w_dc=0.15, prior_init=0.1, min_depth=mind, max_depth=maxd, window_rule=_lib.WINDOW_REFERENCE, depth_param=_lib.DEPTH_QUARTER)
    _, d0, _ = e.refine_dense_window(t["tgt"], t["srcs"], dt4, ds5, t["K"], t["pose"], o0, stats=True, argmin=True)
    want0 = 1.0 / orc.up4(orc.down4(1.0 / f32(w["depth_t"])[0]))
    assert np.abs(d0[0, 0].cpu().numpy().astype(np.float64) / want0 - 1).max() < 1e-5
    # H or W not a multiple of 4, or the PAIR rule: refused
    with pytest.raises(RuntimeError):
        e.refine_dense_window(t["tgt"], t["srcs"], dt4, ds5, t["K"], t["pose"], default_opts(n_iters=1, depth_param=_lib.DEPTH_QUARTER), argmin=True)


@pytest.mark.parametrize("B,S,H,W,mind,maxd", [(1, 1, 240, 320, 0.03, 3.0), (1, 2, 192, 640, 0.06, 2.67), (1, 1, 256, 448, 0.03, 3.0), (1, 3, 128, 416, 0.06, 2.67),
                                                  (2, 2, 48, 160, 0.06, 2.67)])
def test_the_reference_leaf_set_quarter_resolution_target_and_sources(B, S, H, W, mind, maxd, orc):
    """depth_param = TCSFM_DEPTH_QUARTER with free_source_depths: the unknowns are the reference's own leaves -- the quarter-resolution maps of
    the target AND of every source (optimizer.py:194-198: one tensor of S + 1 channels, upsampled x4 every epoch).  Poses, the target map and
    every pixel of every (upsampled) source map follow orc_refine_dense_ref_q_free, decisions replayed"""
    from tightly_coupled_sfm_amd.engine import Engine, default_opts
    from tightly_coupled_sfm_amd import _lib
    n_it = 3
    w = _window(B, S, H, W, seed=31)
    N = 2 * S * B
    e = Engine(H, W, N)
    o = default_opts(n_iters=n_it, w_dc=0.15, prior_init=0.1, min_depth=mind, max_depth=maxd, window_rule=_lib.WINDOW_REFERENCE, lambda_depth=1.0,
                     depth_param=_lib.DEPTH_QUARTER, free_source_depths=1)
    t = {k: _dev(v) for k, v in w.items()}
    dt4, ds5 = t["depth_t"][:, None].contiguous(), t["depth_s"][:, :, None].contiguous()
    e.trace_begin(n_it, N)
    pose, depth, st = e.refine_dense_window(t["tgt"], t["srcs"], dt4, ds5, t["K"], t["pose"], o, stats=True, argmin=True)
    bits, _ = e.trace_end()
    pose = pose.cpu().numpy().astype(np.float64); depth = depth.cpu().numpy().astype(np.float64)
    f32 = lambda a: np.asarray(a, dtype=np.float32).astype(np.float64)
    oo = oracle_opts(n_iters=n_it, w_dc=0.15)
    orc.flip_stats_reset()
    po, do, dso, so = orc.refine_dense_ref_q_free(f32(w["tgt"]), f32(w["srcs"]), f32(w["depth_t"]), f32(w["depth_s"]), f32(w["K"]), f32(w["pose"]), oo, argmin=True,
                                                  w_init=0.1, lambda_depth=1.0, min_depth=mind, max_depth=maxd, bits=bits.reshape(n_it, N, H * W))
    nf, hard = orc.flip_stats(n_it)
    check_dense_ref_flips(nf, hard, N * H * W)
    for m in range(N):
        et = np.linalg.norm(pose[m, :3] - po[m, :3]) / np.linalg.norm(po[m, :3]); er = np.linalg.norm(pose[m, 3:] - po[m, 3:]) / np.linalg.norm(po[m, 3:])
        assert et < 1e-4 and er < 1e-4, (m, et, er)
    for s in range(S):
        assert np.abs(depth[s * B:(s + 1) * B, 0] / do - 1).max() < 1e-4, (s, np.abs(depth[s * B:(s + 1) * B, 0] / do - 1).max())
    src_gpu = depth[S * B:, 0].reshape(S, B, H, W)
    dev = np.sort(np.abs(src_gpu / dso - 1).ravel())
    assert dev[-1] < 1e-4, dev[-6:]
    assert np.all(np.diff(so[:, 0]) < 0), so[:, 0]
    # the source maps moved away from their start (the x4 upsampling of the input's quarter-resolution projection)
    start = 1.0 / orc.up4(orc.down4(1.0 / f32(w["depth_s"])[0, 0]))
    assert np.abs(src_gpu[0, 0] / start - 1).max() > 1e-3
    e.close()


@pytest.mark.parametrize("B,S,H,W", [(2, 2, 48, 160), (3, 1, 24, 40), (1, 3, 48, 160)])
def test_quarter_resolution_batches_and_source_counts(B, S, H, W, orc):
    """the quarter-resolution unknown with several targets per call (the batch normalisers couple them; every target has its own cells,
    records and reduced system) and with one / three sources per target: poses and maps follow the oracle"""
    from tightly_coupled_sfm_amd.engine import Engine, default_opts
    from tightly_coupled_sfm_amd import _lib
    n_it = 2
    w = _window(B, S, H, W, seed=57)
    N = 2 * S * B
    e = Engine(H, W, N)
    o = default_opts(n_iters=n_it, w_dc=0.15, prior_init=0.1, min_depth=0.06, max_depth=2.67, window_rule=_lib.WINDOW_REFERENCE, lambda_depth=1.0,
                     depth_param=_lib.DEPTH_QUARTER)
    t = {k: _dev(v) for k, v in w.items()}
    dt4, ds5 = t["depth_t"][:, None].contiguous(), t["depth_s"][:, :, None].contiguous()
    e.trace_begin(n_it, N)
    pose, depth, _ = e.refine_dense_window(t["tgt"], t["srcs"], dt4, ds5, t["K"], t["pose"], o, stats=True, argmin=True)
    bits, _ = e.trace_end()
    pose = pose.cpu().numpy().astype(np.float64); depth = depth.cpu().numpy().astype(np.float64)
    f32 = lambda a: np.asarray(a, dtype=np.float32).astype(np.float64)
    orc.flip_stats_reset()
    po, do, so, rq = orc.refine_dense_ref_q(f32(w["tgt"]), f32(w["srcs"]), f32(w["depth_t"]), f32(w["depth_s"]), f32(w["K"]), f32(w["pose"]),
                                            oracle_opts(n_iters=n_it, w_dc=0.15), argmin=True, w_init=0.1, lambda_depth=1.0, min_depth=0.06, max_depth=2.67,
                                            bits=bits.reshape(n_it, N, H * W))
    nf, hard = orc.flip_stats(n_it)
    assert hard.sum() == 0, (nf, hard)
    for m in range(N):
        et = np.linalg.norm(pose[m, :3] - po[m, :3]) / np.linalg.norm(po[m, :3]); er = np.linalg.norm(pose[m, 3:] - po[m, 3:]) / np.linalg.norm(po[m, 3:])
        assert et < 1e-4 and er < 1e-4, (m, et, er)
    for s in range(S):
        assert np.abs(depth[s * B:(s + 1) * B, 0] / do - 1).max() < 1e-4, (s, np.abs(depth[s * B:(s + 1) * B, 0] / do - 1).max())
    for b in range(B):          # every target's map is the upsampling of ITS cells
        assert np.abs(depth[b, 0] * orc.up4(rq[b]) - 1).max() < 1e-4


@pytest.mark.parametrize("quarter", [False, True], ids=["full-resolution", "quarter-resolution"])
def test_dense_reference_with_smoothness_term_follows_the_oracle(quarter, orc):
    """l_smooth as a cost term (opts.w_smooth; optimizer.py:92-93): the iterates with the term follow the oracle, with the per-pixel and with
    the quarter-resolution unknown; the term changes the result; outside the reference-loss mode it is refused"""
    from tightly_coupled_sfm_amd.engine import Engine, default_opts
    from tightly_coupled_sfm_amd import _lib
    B, S, H, W, n_it = 2, 2, 48, 160, 3
    w = _window(B, S, H, W, seed=77)
    N = 2 * S * B
    e = Engine(H, W, N)
    kw = dict(n_iters=n_it, w_dc=0.15, prior_init=0.1, min_depth=0.06, max_depth=2.67, window_rule=_lib.WINDOW_REFERENCE, lambda_depth=1.0,
              depth_param=_lib.DEPTH_QUARTER if quarter else _lib.DEPTH_FULL)
    t = {k: _dev(v) for k, v in w.items()}
    dt4, ds5 = t["depth_t"][:, None].contiguous(), t["depth_s"][:, :, None].contiguous()
    e.trace_begin(n_it, N)
    pose, depth, _ = e.refine_dense_window(t["tgt"], t["srcs"], dt4, ds5, t["K"], t["pose"], default_opts(w_smooth=2.0, **kw), stats=True, argmin=True)
    bits, _ = e.trace_end()
    pose = pose.cpu().numpy().astype(np.float64); depth = depth.cpu().numpy().astype(np.float64)
    f32 = lambda a: np.asarray(a, dtype=np.float32).astype(np.float64)
    orc.flip_stats_reset()
    fn = orc.refine_dense_ref_q if quarter else orc.refine_dense_ref
    res = fn(f32(w["tgt"]), f32(w["srcs"]), f32(w["depth_t"]), f32(w["depth_s"]), f32(w["K"]), f32(w["pose"]), oracle_opts(n_iters=n_it, w_dc=0.15, w_smooth=2.0),
             argmin=True, w_init=0.1, lambda_depth=1.0, min_depth=0.06, max_depth=2.67, bits=bits.reshape(n_it, N, H * W))
    po, do, so = res[0], res[1], res[2]
    nf, hard = orc.flip_stats(n_it)
    assert hard.sum() == 0, (nf, hard)
    for m in range(N):
        et = np.linalg.norm(pose[m, :3] - po[m, :3]) / np.linalg.norm(po[m, :3]); er = np.linalg.norm(pose[m, 3:] - po[m, 3:]) / np.linalg.norm(po[m, 3:])
        assert et < 1e-4 and er < 1e-4, (m, et, er)
    assert np.abs(depth[:B, 0] / do - 1).max() < 1e-4, np.abs(depth[:B, 0] / do - 1).max()
    assert np.all(np.diff(so[:, 0]) < 0), so[:, 0]
    _, d_plain, _ = e.refine_dense_window(t["tgt"], t["srcs"], dt4, ds5, t["K"], t["pose"], default_opts(**kw), stats=True, argmin=True)
    assert np.abs(d_plain[:B, 0].cpu().numpy() / depth[:B, 0] - 1).max() > 1e-4          # the term moves the map
    with pytest.raises(RuntimeError):
        e.refine_dense_window(t["tgt"], t["srcs"], dt4, ds5, t["K"], t["pose"], default_opts(n_iters=1, w_smooth=2.0), argmin=True)


@pytest.mark.parametrize("H,W,B", [(48, 160, 2), (192, 640, 1)], ids=["48x160-B2", "192x640-B1"])
def test_lean_joint_kernel_agrees_with_the_rolled_instantiation(H, W, B):
    """Round 5, third session: with two sources and no l_smooth the forward groups run the LEAN form of k_dense_joint (source loop unrolled, Jacobian
    rebuilt in phase 2b, software-pipelined window reads, 168 VGPRs); with l_smooth they run the rolled 256-register instantiation of the same
    source.  A vanishing smoothness weight (1e-30: its terms are far below one ulp of anything they are added to) selects the rolled kernel
    without changing the problem -- the two kernels must then return the same poses and the same map, with no oracle in the loop."""
    from tightly_coupled_sfm_amd.engine import Engine, default_opts
    from tightly_coupled_sfm_amd import _lib
    S, n_it = 2, 4
    w = _window(B, S, H, W, seed=123)
    N = 2 * S * B
    e = Engine(H, W, N)
    t = {k: _dev(v) for k, v in w.items()}
    dt4, ds5 = t["depth_t"][:, None].contiguous(), t["depth_s"][:, :, None].contiguous()
    for quarter in (False, True):
        kw = dict(n_iters=n_it, w_dc=0.15, prior_init=0.1, min_depth=0.06, max_depth=2.67, window_rule=_lib.WINDOW_REFERENCE,
                  depth_param=_lib.DEPTH_QUARTER if quarter else _lib.DEPTH_FULL)
        p_lean, d_lean, s_lean = e.refine_dense_window(t["tgt"], t["srcs"], dt4, ds5, t["K"], t["pose"], default_opts(**kw), stats=True, argmin=True)
        p_roll, d_roll, s_roll = e.refine_dense_window(t["tgt"], t["srcs"], dt4, ds5, t["K"], t["pose"], default_opts(w_smooth=1e-30, **kw), stats=True, argmin=True)
        p_lean, p_roll = p_lean.cpu().numpy().astype(np.float64), p_roll.cpu().numpy().astype(np.float64)
        for m in range(N):
            et = np.linalg.norm(p_lean[m, :3] - p_roll[m, :3]) / np.linalg.norm(p_roll[m, :3]); er = np.linalg.norm(p_lean[m, 3:] - p_roll[m, 3:]) / np.linalg.norm(p_roll[m, 3:])
            assert et < 2e-6 and er < 2e-6, (quarter, m, et, er)
        dl, dr = d_lean[:B, 0].cpu().numpy().astype(np.float64), d_roll[:B, 0].cpu().numpy().astype(np.float64)
        # (a pixel whose mask / selection is an exact tie can fall the other way between two instruction schedules: allow a handful, bound the rest tightly)
        rel = np.abs(dl / dr - 1)
        assert (rel > 1e-5).mean() < 1e-4 and np.median(rel) < 1e-6, ((rel > 1e-5).mean(), np.median(rel), rel.max())
        c_lean, c_roll = s_lean[:, :, 0].cpu().numpy(), s_roll[:, :, 0].cpu().numpy()
        assert np.allclose(c_lean, c_roll, rtol=1e-5, atol=0), (c_lean, c_roll)


@pytest.mark.parametrize("mode", ["full", "quarter", "free"])
def test_dense_reference_with_pose_consistency_term_follows_the_oracle(mode, orc):
    """l_pose_consist as a term of the dense mode (opts.w_pose_consist; optimizer.py:95-96 beside :235-268): 0.1 mean |p_fwd + p_inv| added to the
    reduced pose systems -- the forward pairs' diagonal blocks of the target's joint system, the inverse pairs' own systems (or their groups with
    the source maps free).  The iterates follow the oracle with the term (its weight raised so that it matters), the term changes the poses and
    pulls p_fwd + p_inv towards zero; under the PAIR rule it is refused"""
    from tightly_coupled_sfm_amd.engine import Engine, default_opts
    from tightly_coupled_sfm_amd import _lib
    B, S, H, W, n_it, wpc = 2, 2, 48, 160, 3, 0.5
    w = _window(B, S, H, W, seed=53)
    N = 2 * S * B
    w["pose"] = w["pose"].copy()
    w["pose"][S * B:] += np.array([0.004, -0.003, 0.005, 0.002, 0.003, -0.002])      # the inverse poses start off the negated forward poses: r != 0
    e = Engine(H, W, N)
    kw = dict(n_iters=n_it, w_dc=0.15, prior_init=0.1, min_depth=0.06, max_depth=2.67, window_rule=_lib.WINDOW_REFERENCE, lambda_depth=1.0,
              depth_param=_lib.DEPTH_QUARTER if mode == "quarter" else _lib.DEPTH_FULL, free_source_depths=1 if mode == "free" else 0)
    t = {k: _dev(v) for k, v in w.items()}
    dt4, ds5 = t["depth_t"][:, None].contiguous(), t["depth_s"][:, :, None].contiguous()
    e.trace_begin(n_it, N)
    pose, depth, _ = e.refine_dense_window(t["tgt"], t["srcs"], dt4, ds5, t["K"], t["pose"], default_opts(w_pose_consist=wpc, **kw), stats=True, argmin=True)
    bits, _ = e.trace_end()
    pose = pose.cpu().numpy().astype(np.float64); depth = depth.cpu().numpy().astype(np.float64)
    f32 = lambda a: np.asarray(a, dtype=np.float32).astype(np.float64)
    orc.flip_stats_reset()
    fn = {"full": orc.refine_dense_ref, "quarter": orc.refine_dense_ref_q, "free": orc.refine_dense_ref_free}[mode]
    res = fn(f32(w["tgt"]), f32(w["srcs"]), f32(w["depth_t"]), f32(w["depth_s"]), f32(w["K"]), f32(w["pose"]), oracle_opts(n_iters=n_it, w_dc=0.15, w_pose_consist=wpc),
             argmin=True, w_init=0.1, lambda_depth=1.0, min_depth=0.06, max_depth=2.67, bits=bits.reshape(n_it, N, H * W))
    po, do, so = res[0], res[1], res[-1] if mode == "free" else res[2]
    nf, hard = orc.flip_stats(n_it)
    assert hard.sum() == 0, (nf, hard)
    for m in range(N):
        et = np.linalg.norm(pose[m, :3] - po[m, :3]) / np.linalg.norm(po[m, :3]); er = np.linalg.norm(pose[m, 3:] - po[m, 3:]) / np.linalg.norm(po[m, 3:])
        assert et < 1e-4 and er < 1e-4, (m, et, er)
    assert np.abs(depth[:B, 0] / do - 1).max() < 1e-4, np.abs(depth[:B, 0] / do - 1).max()
    if mode == "free":
        assert np.abs(depth[S * B:, 0].reshape(S, B, H, W) / res[2] - 1).max() < 1e-4
    assert np.all(np.diff(so[:, 0]) < 0), so[:, 0]
    p_plain, _, _ = e.refine_dense_window(t["tgt"], t["srcs"], dt4, ds5, t["K"], t["pose"], default_opts(**kw), stats=True, argmin=True)
    p_plain = p_plain.cpu().numpy().astype(np.float64)
    r_with = np.abs(pose[:S * B] + pose[S * B:]).mean(); r_plain = np.abs(p_plain[:S * B] + p_plain[S * B:]).mean()
    assert np.abs(p_plain - pose).max() > 1e-5 and r_with < r_plain, (r_with, r_plain)      # the term moves the poses, towards p_fwd = -p_inv
    with pytest.raises(RuntimeError):
        e.refine_dense_window(t["tgt"], t["srcs"], dt4, ds5, t["K"], t["pose"], default_opts(n_iters=1, w_pose_consist=wpc), argmin=True)
    e.close()
