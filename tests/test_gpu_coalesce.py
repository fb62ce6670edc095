"""Coalesced calls (tcsfm_refine_window_queued / tcsfm_set_coalesce / tcsfm_flush, round 4): queued B-window calls of one shape run
as ONE launch sequence through a pointer table -- per window the bits of the call run on its own, whatever the grouping."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _calls(n, H, W, S=1, B=1, seed=50):
    from tightly_coupled_sfm_amd import synth
    out = []
    for i in range(n):
        b = synth.make_batch(2 * S * B, H, W, seed0=seed + 13 * i, both_directions=True)
        d = {k: torch.as_tensor(v).cuda().contiguous() for k, v in b.items()}
        # window layout from the pair form: targets = every second pair's target, its source(s) the partner's
        tgt = d["tgt"][0::2][:B].contiguous()
        srcs = torch.stack([d["src"][0::2][s * B:(s + 1) * B] if S > 1 else d["src"][0::2][:B] for s in range(S)]).contiguous()
        dt = d["depth_t"][0::2][:B].contiguous()
        ds = torch.stack([d["depth_s"][0::2][s * B:(s + 1) * B] if S > 1 else d["depth_s"][0::2][:B] for s in range(S)]).contiguous()
        pose = torch.cat([d["pose_init"][0::2][:S * B], d["pose_init"][1::2][:S * B]]).contiguous()
        out.append(dict(tgt=tgt, srcs=srcs, dt=dt, ds=ds, K=d["K"][0::2][:B].contiguous(), pose=pose))
    K0 = out[0]["K"]
    for c in out:
        c["K"] = K0           # one camera
    return out


def test_coalesced_calls_are_bit_identical_to_single_calls():
    from tightly_coupled_sfm_amd import _lib
    from tightly_coupled_sfm_amd.engine import Engine, default_opts
    H, W = 96, 320
    calls = _calls(7, H, W)
    o = default_opts(n_iters=4)
    ref = Engine(H, W, 2)
    want = [ref.refine_window(c["tgt"], c["srcs"], c["dt"], c["ds"], c["K"], c["pose"], o)[0].clone() for c in calls]
    torch.cuda.synchronize()
    e = Engine(H, W, 2 * 4)
    e.set_coalesce(4)
    outs = [torch.zeros_like(c["pose"]) for c in calls]
    for c, out in zip(calls, outs):
        e.refine_window_queued(c["tgt"], c["srcs"], c["dt"], c["ds"], c["K"], c["pose"], out, o)
    assert e.coalesce_counts() == (1, 4)                       # four ran when the queue filled; three are waiting
    e.synchronize()                                            # (flushes)
    assert e.coalesce_counts() == (2, 7)
    for got, w in zip(outs, want):
        assert torch.equal(got, w)
    # another shape / other options flush what is waiting; the REFERENCE rule couples a call's windows and is never merged
    o2 = default_opts(n_iters=2, w_dc=0.15)
    w2 = [ref.refine_window(c["tgt"], c["srcs"], c["dt"], c["ds"], c["K"], c["pose"], o2)[0].clone() for c in calls[:2]]
    orf = default_opts(n_iters=2, window_rule=_lib.WINDOW_REFERENCE)
    wr = ref.refine_window(calls[2]["tgt"], calls[2]["srcs"], calls[2]["dt"], calls[2]["ds"], calls[2]["K"], calls[2]["pose"], orf)[0].clone()
    torch.cuda.synchronize()
    for out in outs:
        out.zero_()
    torch.cuda.synchronize()
    e.refine_window_queued(calls[0]["tgt"], calls[0]["srcs"], calls[0]["dt"], calls[0]["ds"], calls[0]["K"], calls[0]["pose"], outs[0], o)
    e.refine_window_queued(calls[0]["tgt"], calls[0]["srcs"], calls[0]["dt"], calls[0]["ds"], calls[0]["K"], calls[0]["pose"], outs[3], o2)   # flushes the first
    e.refine_window_queued(calls[1]["tgt"], calls[1]["srcs"], calls[1]["dt"], calls[1]["ds"], calls[1]["K"], calls[1]["pose"], outs[4], o2)
    e.refine_window_queued(calls[2]["tgt"], calls[2]["srcs"], calls[2]["dt"], calls[2]["ds"], calls[2]["K"], calls[2]["pose"], outs[5], orf)  # flushes those, runs at once
    e.flush(); e.synchronize()
    assert torch.equal(outs[0], want[0]) and torch.equal(outs[3], w2[0]) and torch.equal(outs[4], w2[1]) and torch.equal(outs[5], wr)
    assert e.coalesce_counts() == (5, 11)
    # coalescing off: every queued call runs at once
    e.set_coalesce(0)
    outs[6].zero_()
    e.refine_window_queued(calls[6]["tgt"], calls[6]["srcs"], calls[6]["dt"], calls[6]["ds"], calls[6]["K"], calls[6]["pose"], outs[6], o)
    assert e.coalesce_counts() == (6, 12)
    e.synchronize()
    assert torch.equal(outs[6], want[6])


def test_coalesced_kitti_windows_two_targets_two_sources():
    """calls with B = 2 targets and S = 2 sources each, min over the sources: the pointer table addresses (call, local target, source)"""
    from tightly_coupled_sfm_amd.engine import Engine, default_opts
    import test_gpu_dense_reference as T
    H, W, B, S = 48, 160, 2, 2
    calls = []
    for i in range(3):
        w = T._window(B, S, H, W, seed=70 + i, bias=1.0)
        t = {k: T._dev(v) for k, v in w.items()}
        calls.append(dict(tgt=t["tgt"], srcs=t["srcs"], dt=t["depth_t"][:, None].contiguous(), ds=t["depth_s"][:, :, None].contiguous(), K=t["K"], pose=t["pose"]))
    for c in calls:
        c["K"] = calls[0]["K"]
    o = default_opts(n_iters=3, w_dc=0.15)
    o.argmin = 1
    ref = Engine(H, W, 2 * S * B)
    want = [ref.refine_window(c["tgt"], c["srcs"], c["dt"], c["ds"], c["K"], c["pose"], o)[0].clone() for c in calls]
    torch.cuda.synchronize()
    e = Engine(H, W, 2 * S * B * 3)
    e.set_coalesce(3)
    outs = [torch.zeros_like(c["pose"]) for c in calls]
    for c, out in zip(calls, outs):
        e.refine_window_queued(c["tgt"], c["srcs"], c["dt"], c["ds"], c["K"], c["pose"], out, o)
    e.synchronize()
    assert e.coalesce_counts() == (1, 3)
    for got, w_ in zip(outs, want):
        assert torch.equal(got, w_)


def test_merged_sequences_alternating_over_lanes():
    """tcsfm_set_coalesce_lanes: consecutive merged sequences run on different streams of the handle (the solve kernels of one overlap
    the launches of the other); every window's poses are still the bits of its own call, tcsfm_flush orders the handle's stream (and a
    consumer queued on it) behind every sequence, and a lane count above the handle's lanes is refused"""
    from tightly_coupled_sfm_amd.engine import Engine, default_opts
    H, W = 96, 320
    calls = _calls(11, H, W, seed=90)
    o = default_opts(n_iters=4)
    ref = Engine(H, W, 2)
    want = [ref.refine_window(c["tgt"], c["srcs"], c["dt"], c["ds"], c["K"], c["pose"], o)[0].clone() for c in calls]
    torch.cuda.synchronize()
    e = Engine(H, W, 2 * 3, lanes=3)
    with pytest.raises(RuntimeError):
        e.set_coalesce_lanes(4)
    e.set_coalesce(3)
    e.set_coalesce_lanes(3)
    for rep in range(3):                                          # sequences 0..3 of every round land on lanes 0, 1, 2, 0 (then rotate on)
        outs = [torch.zeros_like(c["pose"]) for c in calls]
        for c, out in zip(calls, outs):
            e.refine_window_queued(c["tgt"], c["srcs"], c["dt"], c["ds"], c["K"], c["pose"], out, o)
        e.flush()
        # a consumer on the handle's stream (what Engine.set_stream / the current torch stream is NOT: order it explicitly)
        e.synchronize()
        for got, w in zip(outs, want):
            assert torch.equal(got, w)
    assert e.coalesce_counts() == (12, 33)
    # back to one stream, and fewer lanes than coal_lanes: clamped, still the same bits
    e.set_lanes(2)
    outs = [torch.zeros_like(c["pose"]) for c in calls]
    for c, out in zip(calls, outs):
        e.refine_window_queued(c["tgt"], c["srcs"], c["dt"], c["ds"], c["K"], c["pose"], out, o)
    e.synchronize()
    for got, w in zip(outs, want):
        assert torch.equal(got, w)
    e.set_coalesce_lanes(1)
    e.set_coalesce(0)


def test_example_queued_windows_runs():
    """examples/queued_windows.py: the three ways of running many B = 1 windows give the same bits (the example asserts it)"""
    import os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "examples", "queued_windows.py"), "--windows", "60", "--distinct", "12"],
                       capture_output=True, text=True, timeout=600, cwd=root)
    assert r.returncode == 0, r.stderr[-2000:]
    assert "same poses: True" in r.stdout


def test_merged_dense_calls_are_bit_identical_to_single_calls():
    """tcsfm_refine_dense_window_queued: per-pair Gauss-Newton dense calls (one source per target) merged through the pointer table -- every
    call's poses AND depth maps are the bits of the call on its own; dense and pose calls are never merged with each other; a KITTI window
    (S = 2, joint mode) runs at once"""
    from tightly_coupled_sfm_amd.engine import Engine, default_opts
    H, W = 96, 320
    calls = _calls(7, H, W, seed=120)
    o = default_opts(n_iters=3, min_depth=0.03, max_depth=3.0)
    ref = Engine(H, W, 2)
    want = []
    for c in calls:
        p, d, _ = ref.refine_dense_window(c["tgt"], c["srcs"], c["dt"][:, None].contiguous() if c["dt"].dim() == 3 else c["dt"], c["ds"][:, :, None].contiguous() if c["ds"].dim() == 4 else c["ds"],
                                          c["K"], c["pose"], o)
        want.append((p.clone(), d.clone()))
    torch.cuda.synchronize()
    dt4 = [c["dt"][:, None].contiguous() if c["dt"].dim() == 3 else c["dt"] for c in calls]
    ds5 = [c["ds"][:, :, None].contiguous() if c["ds"].dim() == 4 else c["ds"] for c in calls]
    e = Engine(H, W, 2 * 4, lanes=2)
    e.set_coalesce(4); e.set_coalesce_lanes(2)
    po = [torch.zeros(2, 6, device="cuda") for _ in calls]
    do = [torch.zeros(2, 1, H, W, device="cuda") for _ in calls]
    for c, a, b, p, d in zip(calls, dt4, ds5, po, do):
        e.refine_dense_window_queued(c["tgt"], c["srcs"], a, b, c["K"], c["pose"], p, d, o)
    assert e.coalesce_counts() == (1, 4)
    # a pose call of the same shape flushes the waiting dense calls (three) before it is queued itself
    op = default_opts(n_iters=3)
    pose_only = torch.zeros(2, 6, device="cuda")
    e.refine_window_queued(calls[0]["tgt"], calls[0]["srcs"], calls[0]["dt"], calls[0]["ds"], calls[0]["K"], calls[0]["pose"], pose_only, op)
    assert e.coalesce_counts() == (2, 7)
    e.synchronize()
    assert e.coalesce_counts() == (3, 8)
    for (wp, wd), p, d in zip(want, po, do):
        assert torch.equal(p, wp) and torch.equal(d, wd)
    assert torch.equal(pose_only, ref.refine_window(calls[0]["tgt"], calls[0]["srcs"], calls[0]["dt"], calls[0]["ds"], calls[0]["K"], calls[0]["pose"], op)[0])
    e.set_coalesce_lanes(1); e.set_coalesce(0)


@pytest.mark.parametrize("S,B,quarter,H,W", [(1, 1, False, 96, 320), (2, 1, False, 96, 320), (2, 2, True, 48, 160), (1, 1, True, 96, 320), (3, 1, False, 48, 160)])
def test_merged_reference_loss_dense_calls_are_bit_identical_to_single_calls(S, B, quarter, H, W):
    """round 5: queued dense calls under window_rule REFERENCE (the reference's own loss, optimizer.py:47-90) merge as well -- the loss couples
    the windows of ONE call through its batch normalisers, so inside the merged sequence every call is a normaliser group of its own
    (k_linearize<FRONT> counts per group, the joint / solve kernels read their group's counts): poses and depth maps of every call are the
    bits of the call run on its own, with the per-pixel and with the quarter-resolution unknown, for one to three sources"""
    from tightly_coupled_sfm_amd.engine import Engine, default_opts
    from tightly_coupled_sfm_amd import _lib
    calls = _calls(6, H, W, S=S, B=B, seed=300)
    o = default_opts(n_iters=3, w_dc=0.15, prior_init=0.1, min_depth=0.06, max_depth=2.67, window_rule=_lib.WINDOW_REFERENCE,
                     depth_param=_lib.DEPTH_QUARTER if quarter else _lib.DEPTH_FULL)
    o.argmin = 1
    N = 2 * S * B
    ref = Engine(H, W, N)
    dt4 = [c["dt"][:, None].contiguous() if c["dt"].dim() == 3 else c["dt"] for c in calls]
    ds5 = [c["ds"][:, :, None].contiguous() if c["ds"].dim() == 4 else c["ds"] for c in calls]
    want = []
    for c, a, b in zip(calls, dt4, ds5):
        p, d, _ = ref.refine_dense_window(c["tgt"], c["srcs"], a, b, c["K"], c["pose"], o, argmin=True)
        want.append((p.clone(), d.clone()))
    torch.cuda.synchronize()
    e = Engine(H, W, N * 4, lanes=2)
    e.set_coalesce(4); e.set_coalesce_lanes(2)
    po = [torch.zeros(N, 6, device="cuda") for _ in calls]
    do = [torch.zeros(N, 1, H, W, device="cuda") for _ in calls]
    for c, a, b, p, d in zip(calls, dt4, ds5, po, do):
        e.refine_dense_window_queued(c["tgt"], c["srcs"], a, b, c["K"], c["pose"], p, d, o)
    assert e.coalesce_counts() == (1, 4)                 # four calls ran as one sequence, two are waiting
    e.synchronize()
    assert e.coalesce_counts() == (2, 6)
    for i, ((wp, wd), p, d) in enumerate(zip(want, po, do)):
        assert torch.equal(p, wp), (i, (p - wp).abs().max())
        assert torch.equal(d, wd), (i, (d - wd).abs().max())
    # the free-source-map mode is not merged: it runs at once (and still gives the bits of the plain call)
    o2 = default_opts(n_iters=2, w_dc=0.15, prior_init=0.1, min_depth=0.06, max_depth=2.67, window_rule=_lib.WINDOW_REFERENCE, free_source_depths=1)
    o2.argmin = 1
    p2, d2 = torch.zeros(N, 6, device="cuda"), torch.zeros(N, 1, H, W, device="cuda")
    e.refine_dense_window_queued(calls[0]["tgt"], calls[0]["srcs"], dt4[0], ds5[0], calls[0]["K"], calls[0]["pose"], p2, d2, o2)
    assert e.coalesce_counts() == (2, 6)
    wp2, wd2, _ = ref.refine_dense_window(calls[0]["tgt"], calls[0]["srcs"], dt4[0], ds5[0], calls[0]["K"], calls[0]["pose"], o2, argmin=True)
    e.synchronize()
    assert torch.equal(p2, wp2) and torch.equal(d2, wd2)
    e.set_coalesce_lanes(1); e.set_coalesce(0)
    e.close(); ref.close()


def test_merged_pose_scale_calls_are_bit_identical_to_single_calls():
    """tcsfm_refine_window_scale_queued: BASELINE config 4 (pose + one log depth-scale per pair, 7 x 7) through the merged sequences -- poses
    and log scales of every call are the bits of the call on its own, with given and with default (0) initial scales"""
    from tightly_coupled_sfm_amd import _lib
    from tightly_coupled_sfm_amd.engine import Engine, default_opts
    H, W = 96, 320
    calls = _calls(5, H, W, seed=150)
    o = default_opts(n_iters=6, refine=_lib.REFINE_POSE_SCALE)
    ref = Engine(H, W, 2)
    ls0 = [torch.tensor([0.01 * (i - 2), -0.005 * i], device="cuda") for i in range(len(calls))]
    want = [ref.refine_window(c["tgt"], c["srcs"], c["dt"], c["ds"], c["K"], c["pose"], o, log_scale=(l if i % 2 else None))[:2] for i, (c, l) in enumerate(zip(calls, ls0))]
    want = [(p.clone(), l.clone()) for p, l in want]
    torch.cuda.synchronize()
    e = Engine(H, W, 2 * 3, lanes=2)
    e.set_coalesce(3); e.set_coalesce_lanes(2)
    po = [torch.zeros(2, 6, device="cuda") for _ in calls]
    lo = [torch.zeros(2, device="cuda") for _ in calls]
    for i, (c, l, p, q) in enumerate(zip(calls, ls0, po, lo)):
        e.refine_window_scale_queued(c["tgt"], c["srcs"], c["dt"], c["ds"], c["K"], c["pose"], l if i % 2 else None, p, q, o)
    e.synchronize()
    assert e.coalesce_counts() == (2, 5)
    for (wp, wl), p, q in zip(want, po, lo):
        assert torch.equal(p, wp) and torch.equal(q, wl)
    assert float(torch.stack(lo).abs().max()) > 1e-4
    e.set_coalesce_lanes(1); e.set_coalesce(0)


def test_lanes_are_probed_and_safe_when_the_handle_comes_first():
    """a handle created BEFORE the process's first device work used to give lanes that were slower than one stream.
    tcsfm_set_lanes now measures whether the streams overlap and falls back to the handle's own stream when they do not: in a fresh process
    that creates Engine(..., lanes=4) before anything else touches the card, four lanes are never slower than one call in flight"""
    import os, subprocess, sys, json
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = r'''
import json, os, sys, time
sys.path.insert(0, %r)
os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1"); os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import torch
from tightly_coupled_sfm_amd.engine import Engine, default_opts
from tightly_coupled_sfm_amd import synth
H, W = 192, 640
e = Engine(H, W, 2, lanes=4)                      # the handle and its lanes FIRST: no upload, no kernel so far
probe = e.lane_probe()
e.use_own_stream()
ws = []
for j in range(8):
    b = synth.make_batch(2, H, W, seed0=11 * j, both_directions=True)
    d = {k: torch.as_tensor(v).cuda().contiguous() for k, v in b.items()}
    ws.append(dict(tgt=d["tgt"][0::2].contiguous(), srcs=d["src"][0::2].contiguous()[None], dt=d["depth_t"][0::2].contiguous(), ds=d["depth_s"][0::2].contiguous()[None],
                   K=d["K"][0::2].contiguous(), pose=torch.cat([d["pose_init"][0::2], d["pose_init"][1::2]]).contiguous(), out=torch.zeros(2, 6, device="cuda")))
for w in ws: w["K"] = ws[0]["K"]
o = default_opts(n_iters=4)
def run(nl, n):
    for k in range(n):
        w = ws[k %% 8]
        e.refine_window_async(k %% nl, w["tgt"], w["srcs"], w["dt"], w["ds"], w["K"], w["pose"], w["out"], o)
    for l in range(4): e.lane_synchronize(l)
def rate(nl):
    run(nl, 40); torch.cuda.synchronize()
    best = 0.0
    for _ in range(5):
        t0 = time.perf_counter(); run(nl, 200); torch.cuda.synchronize(); best = max(best, 200 / (time.perf_counter() - t0))
    return best
r1 = rate(1); outs1 = [w["out"].clone() for w in ws]
r4 = rate(4); same = all(torch.equal(a, w["out"]) for a, w in zip(outs1, ws))
print(json.dumps({"probe": probe, "one": r1, "four": r4, "same": same}))
''' % root
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300, cwd=root)
    assert r.returncode == 0, r.stderr[-2000:]
    d = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert d["same"] is True
    assert d["probe"]["one_stream_us"] > 0 and d["probe"]["two_streams_us"] > 0
    assert d["four"] >= 0.9 * d["one"], d                      # lanes never lose (they overlap, or the probe switched them off)
    if d["probe"]["serial"]:
        assert "lanes do not overlap" in r.stderr
