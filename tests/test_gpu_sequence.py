"""End to end on a synthetic sequence, the way run_sequential_optimization.py uses the optimiser:
per-window refinement of PoseNet-quality initial poses, then trajectory composition (validate.py:61-68) and the odometry
error metrics.

The yardstick is the photometric optimum, not the scene's ground-truth pose: the reference's warp samples at
u_proj W/(W-1) - 1/2 ("identity pose is not an identity warp"), while the synthetic views are rendered
from the true geometry, so the minimiser of the reference's residual sits a few per cent of the motion away from the
true pose (bench.py reports that distance as `check`).  In the reference's own pipeline the networks are trained through
the same warp and absorb the convention."""
import numpy as np
import pytest

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu


def test_refined_trajectory_beats_initial_trajectory():
    from tightly_coupled_sfm_amd import synth, trajectory
    from tightly_coupled_sfm_amd.engine import Engine, default_opts
    H, W, NF = 96, 320, 12
    rng = np.random.default_rng(5)
    gt, init, pairs = [], [], []
    for i in range(NF):
        pose = np.array([0.002, -0.001, 0.033, 0.001, -0.003, 0.001]) + rng.normal(scale=[3e-4, 3e-4, 2e-3, 5e-4, 1e-3, 5e-4])
        p = synth.make_pair(H, W, seed=200 + i, pose_gt=pose, dtype=np.float32)
        pairs.append(p); gt.append(p["pose_gt"].astype(np.float64))
        init.append(synth.perturb_pose(p["pose_gt"], 300 + i).astype(np.float64))          # PoseNet-quality start
    t = lambda a: torch.as_tensor(np.ascontiguousarray(a, dtype=np.float32)).cuda()
    stack = lambda k: t(np.stack([p[k] for p in pairs]))
    e = Engine(H, W, NF)
    pose, _, st = e.refine(stack("tgt"), stack("src"), stack("depth_t")[:, None], stack("depth_s")[:, None], stack("K"), t(np.stack(init)),
                           default_opts(n_iters=8), stats=True)
    refined = pose.cpu().numpy().astype(np.float64)
    # the optimum: a long refinement started at the true poses
    opt, _, _ = e.refine(stack("tgt"), stack("src"), stack("depth_t")[:, None], stack("depth_s")[:, None], stack("K"), t(np.stack(gt)),
                         default_opts(n_iters=40))
    optimum = opt.cpu().numpy().astype(np.float64)
    traj_opt, _ = trajectory.compose_trajectory(optimum)
    traj_init, _ = trajectory.compose_trajectory(np.stack(init))
    traj_ref, _ = trajectory.compose_trajectory(refined)
    et_i, er_i = trajectory.mean_err(traj_opt, traj_init)
    et_r, er_r = trajectory.mean_err(traj_opt, traj_ref)
    assert et_r < 0.15 * et_i and er_r < 0.15 * er_i, (et_i, et_r, er_i, er_r)      # 8 iterations from the perturbed start reach it
    path = np.sum(np.linalg.norm(optimum[:, :3], axis=1))
    assert np.linalg.norm(traj_ref[-1][:3, 3] - traj_opt[-1][:3, 3]) < 2e-3 * path  # endpoint agreement over the sequence
    seg = trajectory.segment_errors(traj_opt, traj_ref, [0.1, 0.2])
    assert np.all(seg[:, 1] < 0.01)                                                 # < 1 % translation error on every segment length
    assert np.all(st.cpu().numpy()[:, 7, 0] < st.cpu().numpy()[:, 0, 0])


def test_streamed_sequence_lanes_match_plain_calls():
    """SequenceRefiner (frames uploaded once into a device ring, windows round-robin over the engine's lanes, copies on their own
    stream): bit-identical to plain per-window refine_window calls for 1, 2 and 3 lanes, ring wrap-around included"""
    from tightly_coupled_sfm_amd import synth
    from tightly_coupled_sfm_amd.engine import Engine, default_opts
    from tightly_coupled_sfm_amd.streaming import SequenceRefiner
    H, W, T = 48, 160, 26
    seq = synth.make_sequence(T, H, W, seed=3)
    t = lambda a: torch.as_tensor(np.ascontiguousarray(a, dtype=np.float32)).cuda()
    e = Engine(H, W, 2)
    o = default_opts(n_iters=3)
    K = t(seq["K"][None])
    plain = torch.stack([e.refine_window(t(seq["frames"][w:w + 1]), t(seq["frames"][w + 1:w + 2])[None], t(seq["depths"][w:w + 1]),
                                         t(seq["depths"][w + 1:w + 2])[None], K, t(seq["init"][w]), o)[0] for w in range(T - 1)])
    for lanes in (1, 2, 3):
        sr = SequenceRefiner(H, W, sources=1, lanes=lanes, opts=o)
        assert sr.R < T                                        # the ring wraps around several times
        out = sr.run(seq["frames"], seq["depths"], seq["K"], seq["init"])
        assert torch.equal(out, plain), lanes
        out2 = sr.run(seq["frames"], seq["depths"], seq["K"], seq["init"])     # reusable
        assert torch.equal(out2, plain)
        assert torch.equal(sr.run_native(seq["frames"], seq["depths"], seq["K"], seq["init"]), plain.cpu())   # the C++ loop
    assert not torch.equal(plain, t(seq["init"]))             # (and something was refined)


def test_lane_calls_from_pinned_host_memory():
    """tcsfm_refine_window_async with host_ptrs = 2 (pinned host arrays, no synchronisation inside the call): results equal the
    device-pointer call once the lane has been synchronised; host_ptrs = 1 is refused there"""
    import ctypes as C
    from tightly_coupled_sfm_amd import synth
    from tightly_coupled_sfm_amd.engine import Engine, default_opts
    H, W = 48, 160
    seq = synth.make_sequence(4, H, W, seed=9)
    t = lambda a: torch.as_tensor(np.ascontiguousarray(a, dtype=np.float32))
    e = Engine(H, W, 2, lanes=3)
    o = default_opts(n_iters=3)
    ref = [e.refine_window(t(seq["frames"][w:w + 1]).cuda(), t(seq["frames"][w + 1:w + 2])[None].cuda(), t(seq["depths"][w:w + 1]).cuda(),
                           t(seq["depths"][w + 1:w + 2])[None].cuda(), t(seq["K"][None]).cuda(), t(seq["init"][w]).cuda(), o)[0].cpu() for w in range(3)]
    pin = lambda a: t(a).pin_memory()
    host = [dict(tgt=pin(seq["frames"][w:w + 1]), src=pin(seq["frames"][w + 1:w + 2][None]), dt=pin(seq["depths"][w:w + 1]),
                 ds=pin(seq["depths"][w + 1:w + 2][None]), K=pin(seq["K"][None]), p0=pin(seq["init"][w]), out=torch.zeros(2, 6).pin_memory()) for w in range(3)]
    oh = default_opts(n_iters=3, host_ptrs=2)
    P = lambda x: C.c_void_p(x.data_ptr())
    for w, hb in enumerate(host):                   # three calls in flight on three lanes, none of them blocks the host
        rc = e.lib.tcsfm_refine_window_async(e._h, w, C.byref(oh), 1, 1, P(hb["tgt"]), P(hb["src"]), P(hb["dt"]), P(hb["ds"]), P(hb["K"]), P(hb["p0"]),
                                             None, P(hb["out"]), None, None)
        assert rc == 0, e.lib.tcsfm_last_error(e._h)
    for w in range(3):
        e.lane_synchronize(w)
        assert torch.equal(host[w]["out"], ref[w]), w
    bad = default_opts(n_iters=3, host_ptrs=1)
    hb = host[0]
    assert e.lib.tcsfm_refine_window_async(e._h, 1, C.byref(bad), 1, 1, P(hb["tgt"]), P(hb["src"]), P(hb["dt"]), P(hb["ds"]), P(hb["K"]), P(hb["p0"]),
                                           None, P(hb["out"]), None, None) < 0
    assert e.lib.tcsfm_refine_window_async(e._h, 7, C.byref(oh), 1, 1, P(hb["tgt"]), P(hb["src"]), P(hb["dt"]), P(hb["ds"]), P(hb["K"]), P(hb["p0"]),
                                           None, P(hb["out"]), None, None) < 0          # no such lane


def test_dense_windows_on_lanes_match_synchronous_calls():
    """tcsfm_refine_dense_window_async: different windows in flight on three lanes give, bit for bit, what one synchronous
    tcsfm_refine_dense_window call per window gives (each lane owns its scratch; nothing is shared but the inputs)"""
    from tightly_coupled_sfm_amd import synth
    from tightly_coupled_sfm_amd.engine import Engine, default_opts
    H, W, NWIN = 96, 160, 7
    o = default_opts(n_iters=3, min_depth=0.03, max_depth=3.0)
    e = Engine(H, W, 2, lanes=3)
    wins = []
    for w in range(NWIN):
        b = synth.make_batch(2, H, W, seed0=40 + w, both_directions=True)
        d = {k: torch.as_tensor(v).cuda().contiguous() for k, v in b.items()}
        wins.append(dict(tgt=d["tgt"][0:1].contiguous(), srcs=d["src"][0:1].contiguous()[None], dt=d["depth_t"][0:1].contiguous(),
                         ds=d["depth_s"][0:1].contiguous()[None], K=d["K"][0:1].contiguous(),
                         p0=torch.stack([d["pose_init"][0], d["pose_init"][1]]).contiguous()))
    ref = [e.refine_dense_window(x["tgt"], x["srcs"], x["dt"], x["ds"], x["K"], x["p0"], o)[:2] for x in wins]
    torch.cuda.synchronize()
    po = [torch.empty(2, 6, device="cuda") for _ in range(NWIN)]
    do = [torch.empty(2, 1, H, W, device="cuda") for _ in range(NWIN)]
    for w, x in enumerate(wins):
        e.refine_dense_window_async(w % 3, x["tgt"], x["srcs"], x["dt"], x["ds"], x["K"], x["p0"], po[w], do[w], o)
    for l in range(3):
        e.lane_synchronize(l)
    for w in range(NWIN):
        assert torch.equal(po[w], ref[w][0]) and torch.equal(do[w], ref[w][1]), w
    assert not torch.equal(po[0], po[1])                      # the windows really differ


@pytest.mark.parametrize("S,lanes,ring", [(1, 1, 0), (1, 3, 0), (1, 2, 4), (1, 3, 12), (2, 3, 0), (2, 2, 5), (2, 2, 12)])
def test_native_sequence_loop_matches_per_window_calls(S, lanes, ring):
    """tcsfm_refine_sequence (the reference's whole window loop inside the library: frames uploaded once into a device ring on a
    copy stream, windows round robin on the lanes, slots recycled by events): bit-identical to one refine_window call per window,
    for one and two sources per window, rings that wrap many times (single-frame copies down to the minimum S + 2 slots, four-frame
    copies from S + 8 slots up), pinned and pageable memory"""
    from tightly_coupled_sfm_amd import synth
    from tightly_coupled_sfm_amd.engine import Engine, default_opts
    H, W, T = 48, 160, 23
    seq = synth.make_sequence(T, H, W, seed=11)
    t = lambda a: torch.as_tensor(np.ascontiguousarray(a, dtype=np.float32))
    nwin = T - S
    # initial poses [nwin, 2S, 6]: forward pairs (target -> source k), then inverse pairs; two frames ahead ~ the sum of the steps
    fwd = [[seq["init"][w + k, 0] if k == 0 else seq["init"][w, 0] + seq["init"][w + 1, 0] for k in range(S)] for w in range(nwin)]
    init = np.stack([np.stack(f + [synth.invert_pose(p) for p in f]) for f in fwd]).astype(np.float32)
    o = default_opts(n_iters=3, argmin=1)
    e = Engine(H, W, 2 * S, lanes=lanes)
    K = t(seq["K"][None]).cuda()
    plain = []
    for w in range(nwin):
        srcs = t(seq["frames"][w + 1:w + 1 + S])[:, None].cuda(); ds = t(seq["depths"][w + 1:w + 1 + S])[:, None].cuda()
        plain.append(e.refine_window(t(seq["frames"][w:w + 1]).cuda(), srcs, t(seq["depths"][w:w + 1]).cuda(), ds, K, t(init[w]).cuda(), o)[0].cpu())
    plain = torch.stack(plain)
    frames, depths = t(seq["frames"]).pin_memory(), t(seq["depths"]).pin_memory()
    out = e.refine_sequence(frames, depths, seq["K"], init, o, sources=S, ring=ring)
    assert out.shape == (nwin, 2 * S, 6) and torch.equal(out, plain)
    out2 = e.refine_sequence(t(seq["frames"]), t(seq["depths"]), seq["K"], init, o, sources=S, ring=ring)      # pageable memory, scratch reused
    assert torch.equal(out2, plain)
    assert not torch.equal(plain, t(init))
    # the handle is still good for ordinary calls afterwards, and errors are reported as codes
    again = e.refine_window(t(seq["frames"][0:1]).cuda(), t(seq["frames"][1:1 + S])[:, None].cuda(), t(seq["depths"][0:1]).cuda(),
                            t(seq["depths"][1:1 + S])[:, None].cuda(), K, t(init[0]).cuda(), o)[0].cpu()
    assert torch.equal(again, plain[0])
    with pytest.raises(RuntimeError):
        e.refine_sequence(frames, depths, seq["K"], init, o, sources=S, ring=S + 1)        # ring too small
    badK = seq["K"].copy(); badK[0, 1] = 0.1
    with pytest.raises(RuntimeError, match="pinhole"):
        e.refine_sequence(frames, depths, badK, init, o, sources=S)


@pytest.mark.parametrize("wpc,lanes,ring", [(0, 2, 0), (8, 3, 0), (3, 2, 0), (5, 1, 12), (4, 3, 20), (8, 2, 16), (8, 2, 6), (0, 1, 3)])
def test_sequence_loop_with_several_windows_per_call(wpc, lanes, ring):
    """tcsfm_refine_sequence with one source per window refines `windows_per_call` consecutive windows per call (their targets and
    sources are contiguous runs of the ring): bit-identical to one refine_window call per window for every batch size (ragged last
    call included), lane count and ring -- the refinement is batch-independent"""
    from tightly_coupled_sfm_amd import synth
    from tightly_coupled_sfm_amd.engine import Engine, default_opts
    H, W, T = 48, 160, 31
    seq = synth.make_sequence(T, H, W, seed=12)
    t = lambda a: torch.as_tensor(np.ascontiguousarray(a, dtype=np.float32))
    o = default_opts(n_iters=3, refine=1)                               # pose + log depth scale: the scale array is permuted back too
    e = Engine(H, W, 16, lanes=lanes)
    K = t(seq["K"][None]).cuda()
    plain, plain_ls = [], []
    for w in range(T - 1):
        p, ls, _ = e.refine_window(t(seq["frames"][w:w + 1]).cuda(), t(seq["frames"][w + 1:w + 2])[None].cuda(), t(seq["depths"][w:w + 1]).cuda(),
                                   t(seq["depths"][w + 1:w + 2])[None].cuda(), K, t(seq["init"][w]).cuda(), o)
        plain.append(p.cpu()); plain_ls.append(ls.cpu())
    out, ls = e.refine_sequence(t(seq["frames"]).pin_memory(), t(seq["depths"]).pin_memory(), seq["K"], seq["init"], o, ring=ring,
                                windows_per_call=wpc, log_scale=True)
    assert torch.equal(out, torch.stack(plain)) and torch.equal(ls, torch.stack(plain_ls))


@pytest.mark.parametrize("S,tp,wpc,lanes,ring", [(2, -1, 4, 2, 0), (2, -1, 1, 3, 5), (1, -1, 8, 2, 0), (3, 2, 2, 2, 12), (2, 0, 4, 1, 16), (2, 2, 3, 2, 0)])
def test_sequence_windows_with_the_target_anywhere(S, tp, wpc, lanes, ring):
    """Windows as the reference's loaders form them (data/kitti_loader.py:271-273: S + 1 consecutive frames, the target is the middle
    one -- target_pos = -1 -- and the sources are the others in order), or with the target at any position; several windows per call
    for any number of sources (the targets and every source of consecutive windows are runs of the frame ring, addressed by
    position).  Bit-identical to one refine_window call per window with the frames gathered by hand."""
    from tightly_coupled_sfm_amd import synth
    from tightly_coupled_sfm_amd.engine import Engine, default_opts
    H, W, T = 48, 160, 21
    seq = synth.make_sequence(T, H, W, seed=14)
    t = lambda a: torch.as_tensor(np.ascontiguousarray(a, dtype=np.float32))
    nwin = T - S
    tpos = (S + 1) // 2 if tp < 0 else tp
    src_pos = [k for k in range(S + 1) if k != tpos]
    step = seq["init"][:, 0]                                            # PoseNet-quality pose of frame k -> k+1
    def chain(a, b):                                                    # rough start for frame a -> frame b: sum of the steps (sign by direction)
        return sum(step[k] for k in range(a, b)) if b > a else -sum(step[k] for k in range(b, a))
    init = np.stack([np.stack([chain(w + tpos, w + p) for p in src_pos] + [chain(w + p, w + tpos) for p in src_pos]) for w in range(nwin)]).astype(np.float32)
    o = default_opts(n_iters=3, argmin=1, w_dc=0.15)
    e = Engine(H, W, 2 * S * max(wpc, 1), lanes=lanes)
    K = t(seq["K"][None]).cuda()
    plain = []
    for w in range(nwin):
        srcs = torch.stack([t(seq["frames"][w + p]) for p in src_pos])[:, None].cuda(); ds = torch.stack([t(seq["depths"][w + p]) for p in src_pos])[:, None].cuda()
        plain.append(e.refine_window(t(seq["frames"][w + tpos][None]).cuda(), srcs, t(seq["depths"][w + tpos][None]).cuda(), ds, K, t(init[w]).cuda(), o)[0].cpu())
    out = e.refine_sequence(t(seq["frames"]).pin_memory(), t(seq["depths"]).pin_memory(), seq["K"], init, o, sources=S, ring=ring,
                            windows_per_call=wpc, target_pos=tp)
    assert torch.equal(out, torch.stack(plain))
    assert not torch.equal(out, t(init))


@pytest.mark.parametrize("S,tp,wpc,lanes", [(1, 0, 1, 2), (1, 0, 4, 2), (1, -1, 8, 1), (2, -1, 2, 2)])
def test_dense_sequence_matches_per_window_calls(S, tp, wpc, lanes):
    """tcsfm_refine_dense_sequence (pose + per-pixel inverse depth over a sequence, windows batched per call on the lanes): poses and
    every pair's refined depth map equal, bit for bit, one refine_dense_window call per window"""
    from tightly_coupled_sfm_amd import synth
    from tightly_coupled_sfm_amd.engine import Engine, default_opts
    H, W, T = 48, 96, 14
    seq = synth.make_sequence(T, H, W, seed=15)
    t = lambda a: torch.as_tensor(np.ascontiguousarray(a, dtype=np.float32))
    nwin = T - S
    tpos = (S + 1) // 2 if tp < 0 else tp
    src_pos = [k for k in range(S + 1) if k != tpos]
    step = seq["init"][:, 0]
    chain = lambda a, b: sum(step[k] for k in range(a, b)) if b > a else -sum(step[k] for k in range(b, a))
    init = np.stack([np.stack([chain(w + tpos, w + p) for p in src_pos] + [chain(w + p, w + tpos) for p in src_pos]) for w in range(nwin)]).astype(np.float32)
    o = default_opts(n_iters=3, min_depth=0.03, max_depth=3.0, argmin=1)
    e = Engine(H, W, 2 * S * wpc, lanes=lanes)
    K = t(seq["K"][None]).cuda()
    ref_p, ref_d = [], []
    for w in range(nwin):
        srcs = torch.stack([t(seq["frames"][w + p]) for p in src_pos])[:, None].cuda(); ds = torch.stack([t(seq["depths"][w + p]) for p in src_pos])[:, None].cuda()
        p, d, _ = e.refine_dense_window(t(seq["frames"][w + tpos][None]).cuda(), srcs, t(seq["depths"][w + tpos][None]).cuda(), ds, K, t(init[w]).cuda(), o)
        ref_p.append(p.cpu()); ref_d.append(d.cpu())
    out, dout = e.refine_dense_sequence(t(seq["frames"]).pin_memory(), t(seq["depths"]).pin_memory(), seq["K"], init, o, sources=S,
                                        windows_per_call=wpc, target_pos=tp)
    assert torch.equal(out, torch.stack(ref_p)) and torch.equal(dout, torch.stack(ref_d))
    assert not torch.equal(dout[:, 0], t(seq["depths"][tpos:tpos + nwin]))          # the depth maps were refined
