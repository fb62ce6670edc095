"""tcsfm_set_graph_replay: a repeated device-pointer refine call is captured once and replayed as one HIP graph -- same kernels,
bit-identical results, whatever the buffers hold at replay time; everything else falls back to plain launches."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _batch(H, W, seed):
    from tightly_coupled_sfm_amd import synth
    b = synth.make_batch(2, H, W, seed0=seed, both_directions=True)
    return {k: torch.as_tensor(v).cuda().contiguous() for k, v in b.items()}


def test_replay_is_bit_identical_and_counts():
    from tightly_coupled_sfm_amd.engine import Engine, default_opts
    H, W = 96, 320
    d = _batch(H, W, 3)
    o = default_opts(n_iters=4)
    plain = Engine(H, W, 2)
    want, _, _ = plain.refine(d["tgt"], d["src"], d["depth_t"], d["depth_s"], d["K"], d["pose_init"], o)
    torch.cuda.synchronize()
    e = Engine(H, W, 2, lanes=2)
    assert e.lanes == 2
    e.use_own_stream()
    e.set_graph_replay(2)
    win = dict(tgt=d["tgt"][0:1].contiguous(), srcs=d["src"][0:1].contiguous()[None], depth_t=d["depth_t"][0:1].contiguous(),
               depth_s=d["depth_s"][0:1].contiguous()[None], K=d["K"][0:1].contiguous(), pose=d["pose_init"].clone())
    outs = [torch.zeros_like(win["pose"]) for _ in range(2)]
    for rep in range(5):
        for lane in range(2):
            outs[lane].zero_()
        torch.cuda.synchronize()                        # (the engine runs on its own non-blocking streams: order torch's work by hand)
        for lane in range(2):
            e.refine_window_async(lane, win["tgt"], win["srcs"], win["depth_t"], win["depth_s"], win["K"], win["pose"], outs[lane], o)
        for lane in range(2):
            e.lane_synchronize(lane)
            assert torch.equal(outs[lane], want), (rep, lane)
    assert e.graph_replay_counts() == (2, 6)            # per lane: plain, capture (+ launch), then three replays
    # the graph reads the buffers, not a snapshot: new contents in the same tensors give the new problem's result
    d2 = _batch(H, W, 11)
    want2, _, _ = plain.refine(d2["tgt"], d2["src"], d2["depth_t"], d2["depth_s"], d2["K"], d2["pose_init"], o)
    win["tgt"].copy_(d2["tgt"][0:1]); win["srcs"].copy_(d2["src"][0:1][None]); win["depth_t"].copy_(d2["depth_t"][0:1])
    win["depth_s"].copy_(d2["depth_s"][0:1][None]); win["pose"].copy_(d2["pose_init"])
    torch.cuda.synchronize()
    e.refine_window_async(1, win["tgt"], win["srcs"], win["depth_t"], win["depth_s"], win["K"], win["pose"], outs[1], o)
    e.lane_synchronize(1)
    assert torch.equal(outs[1], want2) and e.graph_replay_counts() == (2, 7)
    # other options = another call: seen for the first time, launched plainly
    o8 = default_opts(n_iters=2)
    w8, _, _ = plain.refine(d2["tgt"], d2["src"], d2["depth_t"], d2["depth_s"], d2["K"], d2["pose_init"], o8)
    e.refine_window_async(1, win["tgt"], win["srcs"], win["depth_t"], win["depth_s"], win["K"], win["pose"], outs[1], o8)
    e.lane_synchronize(1)
    assert torch.equal(outs[1], w8) and e.graph_replay_counts() == (2, 7)
    # switched off: captures dropped, plain launches, same bits
    e.set_graph_replay(0)
    e.refine_window_async(0, win["tgt"], win["srcs"], win["depth_t"], win["depth_s"], win["K"], win["pose"], outs[0], o)
    e.lane_synchronize(0)
    assert torch.equal(outs[0], want2) and e.graph_replay_counts() == (2, 7)


def test_replay_eviction_pair_form_lm_and_bypasses():
    from tightly_coupled_sfm_amd.engine import Engine, default_opts
    H, W = 48, 160
    a, b = _batch(H, W, 1), _batch(H, W, 2)
    torch.cuda.synchronize()
    e = Engine(H, W, 2)
    e.use_own_stream()
    e.set_graph_replay(1)                               # one slot: two alternating calls evict each other, never replayed
    o = default_opts(n_iters=3, solver=1, lambda0=1e-3, w_dc=0.15)
    ref = Engine(H, W, 2)
    wa, _, _ = ref.refine(a["tgt"], a["src"], a["depth_t"], a["depth_s"], a["K"], a["pose_init"], o)
    wb, _, _ = ref.refine(b["tgt"], b["src"], b["depth_t"], b["depth_s"], b["K"], b["pose_init"], o)
    oa, ob = torch.empty_like(wa), torch.empty_like(wb)
    for _ in range(3):
        e.refine_into(a["tgt"], a["src"], a["depth_t"], a["depth_s"], a["K"], a["pose_init"], oa, o)
        e.refine_into(b["tgt"], b["src"], b["depth_t"], b["depth_s"], b["K"], b["pose_init"], ob, o)
    torch.cuda.synchronize()
    assert torch.equal(oa, wa) and torch.equal(ob, wb) and e.graph_replay_counts() == (0, 0)
    e.set_graph_replay(2)                               # two slots: both captured (LM: six launches + the cost-only pass), both replayed
    for _ in range(4):
        e.refine_into(a["tgt"], a["src"], a["depth_t"], a["depth_s"], a["K"], a["pose_init"], oa, o)
        e.refine_into(b["tgt"], b["src"], b["depth_t"], b["depth_s"], b["K"], b["pose_init"], ob, o)
    torch.cuda.synchronize()
    assert torch.equal(oa, wa) and torch.equal(ob, wb) and e.graph_replay_counts() == (2, 5)      # (b was already seen once above)
    # profiling brackets need their own launches: bypass while profiling, replay again afterwards
    e.profile_begin()
    e.refine_into(a["tgt"], a["src"], a["depth_t"], a["depth_s"], a["K"], a["pose_init"], oa, o)
    pr = e.profile_end()
    assert pr["linearize"][1] == 4 and e.graph_replay_counts() == (2, 5)            # 3 linearisations + the cost-only pass, bracketed one by one
    e.refine_into(a["tgt"], a["src"], a["depth_t"], a["depth_s"], a["K"], a["pose_init"], oa, o)
    torch.cuda.synchronize()
    assert torch.equal(oa, wa) and e.graph_replay_counts() == (2, 6)
    # host pointers are never captured
    oh = default_opts(n_iters=3, solver=1, lambda0=1e-3, w_dc=0.15, host_ptrs=1)
    hp = {k: v.cpu().numpy() for k, v in a.items()}
    out = np.empty((2, 6), np.float32)
    import ctypes as C
    P = lambda x: x.ctypes.data_as(C.c_void_p)
    for _ in range(3):
        rc = e.lib.tcsfm_refine(e._h, C.byref(oh), 2, P(hp["tgt"]), P(hp["src"]), P(hp["depth_t"]), P(hp["depth_s"]), P(hp["K"]), P(hp["pose_init"]), None,
                                P(out), None, None)
        assert rc == 0
    assert np.array_equal(out, wa.cpu().numpy()) and e.graph_replay_counts() == (2, 6)


def test_replay_of_dense_calls_per_pair_and_joint():
    """the dense entry points go through the same capture: S = 1 windows on the lanes, and the joint S = 2 window whose inverse pairs run
    on a second stream inside the call (fork / join by events, captured with it)"""
    from tightly_coupled_sfm_amd import synth
    from tightly_coupled_sfm_amd.engine import Engine, default_opts
    H, W = 96, 320
    d = _batch(H, W, 5)
    o = default_opts(n_iters=3, min_depth=0.03, max_depth=3.0)
    tgt, srcs = d["tgt"][0:1].contiguous(), d["src"][0:1].contiguous()[None]
    dt, ds, K, p0 = d["depth_t"][0:1].contiguous(), d["depth_s"][0:1].contiguous()[None], d["K"][0:1].contiguous(), d["pose_init"].clone()
    torch.cuda.synchronize()
    ref = Engine(H, W, 2)
    wp, wd, _ = ref.refine_dense_window(tgt, srcs, dt, ds, K, p0, o)
    e = Engine(H, W, 2, lanes=2)
    e.use_own_stream()
    e.set_graph_replay(2)
    po = [torch.empty(2, 6, device="cuda") for _ in range(2)]
    do = [torch.empty(2, 1, H, W, device="cuda") for _ in range(2)]
    torch.cuda.synchronize()
    for rep in range(4):
        for lane in range(2):
            e.refine_dense_window_async(lane, tgt, srcs, dt, ds, K, p0, po[lane], do[lane], o)
        for lane in range(2):
            e.lane_synchronize(lane)
            assert torch.equal(po[lane], wp) and torch.equal(do[lane], wd), (rep, lane)
    assert e.graph_replay_counts() == (2, 4)
    # joint dense window, S = 2, min over the sources
    S = 2
    b = synth.make_batch(2 * S, H, W, seed0=21)
    t = {k: torch.as_tensor(v).cuda().contiguous() for k, v in b.items()}
    tg2, sr2 = t["tgt"][:1].contiguous(), t["src"][:S].reshape(S, 1, 3, H, W).contiguous()
    dt2, ds2 = t["depth_t"][:1].contiguous(), t["depth_s"][:S].reshape(S, 1, 1, H, W).contiguous()
    pose = torch.cat([t["pose_init"][:S], -t["pose_init"][:S]]).contiguous()
    K2 = t["K"][:1].contiguous()
    torch.cuda.synchronize()
    oj = default_opts(n_iters=3, min_depth=0.03, max_depth=3.0, dense_joint=1)
    ref2 = Engine(H, W, 2 * S)
    jp, jd, _ = ref2.refine_dense_window(tg2, sr2, dt2, ds2, K2, pose, oj, argmin=True)
    e2 = Engine(H, W, 2 * S)
    e2.use_own_stream()
    e2.set_graph_replay(1)
    import ctypes as C
    oj.argmin = 1
    gp, gd = torch.empty_like(jp), torch.empty_like(jd)
    torch.cuda.synchronize()
    for rep in range(4):
        gp.zero_(); gd.zero_()
        torch.cuda.synchronize()
        rc = e2.lib.tcsfm_refine_dense_window(e2._h, C.byref(oj), 1, S, e2._p(tg2), e2._p(sr2), e2._p(dt2), e2._p(ds2), e2._p(K2), e2._p(pose),
                                              e2._p(gp), e2._p(gd), None)
        assert rc == 0
        e2.synchronize()
        assert torch.equal(gp, jp) and torch.equal(gd, jd), rep
    assert e2.graph_replay_counts() == (1, 2)
    # the reference's loss with the reference's complete leaf set (quarter-resolution maps of target AND sources: free_source_depths) through
    # the same capture -- its memsets, the scatter with integer atomics and the inverse groups' launches replay to the same bits
    from tightly_coupled_sfm_amd import _lib
    of = default_opts(n_iters=2, w_dc=0.15, prior_init=0.1, min_depth=0.03, max_depth=3.0, window_rule=_lib.WINDOW_REFERENCE, depth_param=_lib.DEPTH_QUARTER,
                      free_source_depths=1, argmin=1)
    fp, fd, _ = ref2.refine_dense_window(tg2, sr2, dt2, ds2, K2, pose, of, argmin=True)
    torch.cuda.synchronize()
    c0 = e2.graph_replay_counts()
    for rep in range(4):
        gp.zero_(); gd.zero_()
        torch.cuda.synchronize()
        rc = e2.lib.tcsfm_refine_dense_window(e2._h, C.byref(of), 1, S, e2._p(tg2), e2._p(sr2), e2._p(dt2), e2._p(ds2), e2._p(K2), e2._p(pose),
                                              e2._p(gp), e2._p(gd), None)
        assert rc == 0
        e2.synchronize()
        assert torch.equal(gp, fp) and torch.equal(gd, fd), rep
    c1 = e2.graph_replay_counts()
    assert c1[1] - c0[1] == 2 and not torch.equal(fd[S:], ds2.reshape(S, 1, H, W))          # replayed twice; the source maps moved


def test_replay_survives_a_larger_joint_call_and_a_stream_switch():
    """the joint scratch a captured graph points into must not move when a call with more sources arrives, and a handle
    that switches streams must not replay a graph that may still run on the old one: replay S = 2, call S = 3, replay S = 2 again,
    switch the stream, call once more -- every result equals the plain launches' bits."""
    import ctypes as C
    from tightly_coupled_sfm_amd import synth
    from tightly_coupled_sfm_amd.engine import Engine, default_opts
    H, W = 96, 320
    oj = default_opts(n_iters=3, min_depth=0.03, max_depth=3.0, dense_joint=1)
    oj.argmin = 1

    def window(S, seed):
        b = synth.make_batch(2 * S, H, W, seed0=seed)
        t = {k: torch.as_tensor(v).cuda().contiguous() for k, v in b.items()}
        return dict(tgt=t["tgt"][:1].contiguous(), srcs=t["src"][:S].reshape(S, 1, 3, H, W).contiguous(), dt=t["depth_t"][:1].contiguous(),
                    ds=t["depth_s"][:S].reshape(S, 1, 1, H, W).contiguous(), K=t["K"][:1].contiguous(),
                    pose=torch.cat([t["pose_init"][:S], -t["pose_init"][:S]]).contiguous(), S=S)

    w2, w3 = window(2, 21), window(3, 33)
    torch.cuda.synchronize()
    ref = Engine(H, W, 6)
    want = {}
    for w in (w2, w3):
        want[w["S"]] = ref.refine_dense_window(w["tgt"], w["srcs"], w["dt"], w["ds"], w["K"], w["pose"], oj, argmin=True)[:2]
    torch.cuda.synchronize()
    e = Engine(H, W, 6)
    e.use_own_stream()
    e.set_graph_replay(2)

    def call(w):
        gp = torch.zeros(2 * w["S"], 6, device="cuda"); gd = torch.zeros(2 * w["S"], 1, H, W, device="cuda")
        torch.cuda.synchronize()
        rc = e.lib.tcsfm_refine_dense_window(e._h, C.byref(oj), 1, w["S"], e._p(w["tgt"]), e._p(w["srcs"]), e._p(w["dt"]), e._p(w["ds"]), e._p(w["K"]),
                                             e._p(w["pose"]), e._p(gp), e._p(gd), None)
        assert rc == 0, e.last_error()
        e.synchronize()
        assert torch.equal(gp, want[w["S"]][0]) and torch.equal(gd, want[w["S"]][1]), w["S"]

    # the same output tensors must be reused for a call to repeat: keep them per window
    outs = {}
    def call_fixed(w):
        if w["S"] not in outs:
            outs[w["S"]] = (torch.zeros(2 * w["S"], 6, device="cuda"), torch.zeros(2 * w["S"], 1, H, W, device="cuda"))
        gp, gd = outs[w["S"]]
        gp.zero_(); gd.zero_()
        torch.cuda.synchronize()
        rc = e.lib.tcsfm_refine_dense_window(e._h, C.byref(oj), 1, w["S"], e._p(w["tgt"]), e._p(w["srcs"]), e._p(w["dt"]), e._p(w["ds"]), e._p(w["K"]),
                                             e._p(w["pose"]), e._p(gp), e._p(gd), None)
        assert rc == 0, e.last_error()
        e.synchronize()
        assert torch.equal(gp, want[w["S"]][0]) and torch.equal(gd, want[w["S"]][1]), w["S"]

    for _ in range(3):
        call_fixed(w2)                                   # plain, capture, replay
    assert e.graph_replay_counts() == (1, 1)
    call_fixed(w3)                                       # more sources: the joint scratch must stay where the S = 2 graph expects it
    call_fixed(w2)
    assert e.graph_replay_counts() == (1, 2)
    call_fixed(w3); call_fixed(w3)                       # S = 3 captured and replayed beside it
    call_fixed(w2)
    assert e.graph_replay_counts() == (2, 4)
    # stream switch: the captures are dropped (drained first), the next calls run plainly on the new stream, then capture again
    st = torch.cuda.Stream()
    e.set_stream(st.cuda_stream)
    assert e.graph_replay_counts() == (2, 4)
    for _ in range(3):
        call_fixed(w2)
    assert e.graph_replay_counts() == (3, 5)
    call(w3)
