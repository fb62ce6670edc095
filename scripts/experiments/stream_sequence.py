"""PCIe-inclusive throughput of a streamed synthetic sequence: 200 frames of 640x192 in pinned host memory,
window = (frame t, frame t+1) -> fwd + inv directed pairs, 4 GN iterations.  Compares
  (a) the round-1 host-pointer path: one synchronous tcsfm_refine per window, both pairs' arrays handed over (7.9 MB / window)
  (b) SequenceRefiner: every frame uploaded once, window form, 1 / 2 / 3 lanes"""
import json, os, sys, time, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import os as _os; _os.environ.setdefault("TCSFM_SET_ENV_DEFAULTS", "1")      # (a measurement script owns its process: HIP_FORCE_DEV_KERNARG / GPU_MAX_HW_QUEUES when absent)
from tightly_coupled_sfm_amd import synth
from tightly_coupled_sfm_amd.engine import Engine, default_opts
from tightly_coupled_sfm_amd.streaming import SequenceRefiner

H, W, T = 192, 640, int(sys.argv[1]) if len(sys.argv) > 1 else 200
seq = synth.make_sequence(T, H, W, seed=5)
frames = torch.as_tensor(seq["frames"]).pin_memory(); depths = torch.as_tensor(seq["depths"]).pin_memory()
K, init = seq["K"], seq["init"]            # init [T-1, 2, 6]
opts = default_opts(n_iters=4)

# (a) synchronous host-pointer calls, pinned arrays, pair form (what round 1 measured)
e = Engine(H, W, 2)
o = default_opts(n_iters=4, host_ptrs=1)
ptr = lambda a: C.c_void_p(a.data_ptr())
out = torch.zeros((2, 6)).pin_memory()
Kh = torch.as_tensor(np.repeat(K[None], 2, 0).astype(np.float32)).pin_memory()
def window_a(w):
    tg = torch.stack([frames[w], frames[w + 1]]); sr = torch.stack([frames[w + 1], frames[w]])
    dt = torch.stack([depths[w], depths[w + 1]]); ds = torch.stack([depths[w + 1], depths[w]])
    return [x.pin_memory() for x in (tg, sr, dt, ds)]
wins = [window_a(w) for w in range(min(T - 1, 40))]
p_h = torch.as_tensor(init[:40]).pin_memory()
for rep in range(2):
    t0 = time.perf_counter()
    for w, (tg, sr, dt, ds) in enumerate(wins):
        e._call(e.lib.tcsfm_refine(e._h, C.byref(o), 2, ptr(tg), ptr(sr), ptr(dt), ptr(ds), ptr(Kh), ptr(p_h[w]), None, ptr(out), None, None))
    dt_a = (time.perf_counter() - t0) / len(wins)
print(json.dumps({"path": "r01 host-pointer call per window (pinned, synchronous, pair form: 7.9 MB in per window)", "us_per_window": round(dt_a * 1e6, 1),
                  "windows_per_s": round(1 / dt_a, 1)}), flush=True)
e.close()

ref = None
for lanes in (1, 2, 3, 4):
    sr = SequenceRefiner(H, W, sources=1, lanes=lanes, opts=opts)
    sr.run(frames[:20], depths[:20], K, init[:19])            # warm-up (allocations, clocks)
    torch.cuda.synchronize()
    times = []
    for rep in range(7):
        t0 = time.perf_counter()
        res = sr.run(frames, depths, K, init)
        torch.cuda.synchronize()
        times.append(time.perf_counter() - t0)
    best = sorted(times)[len(times) // 2]              # median of 7 passes over the sequence
    if ref is None:
        ref = res.clone()
    same = bool(torch.equal(res, ref))
    print(json.dumps({"path": f"SequenceRefiner: each frame uploaded once ({(3 + 1) * H * W * 4 / 1e6:.2f} MB / window), window form, {lanes} lane(s)",
                      "frames": T, "windows": T - 1, "ms_total": round(best * 1e3, 2), "windows_per_s": round((T - 1) / best, 1),
                      "bit_identical_to_1_lane": same}), flush=True)
    sr.eng.close(); del sr                 # live handles keep their streams: the GPU runs four hardware queues at once
# (c) the same loop inside the library: ONE tcsfm_refine_sequence call per pass (C++ drives copies, events and lanes)
init_t = torch.as_tensor(init)
for lanes, wpc in ((1, 1), (2, 1), (3, 1), (1, 8), (2, 8), (3, 8), (2, 16)):
    e = Engine(H, W, 2 * wpc, lanes=lanes)
    e.refine_sequence(frames[:60], depths[:60], K, init_t[:59], opts, windows_per_call=wpc)
    times = []
    for rep in range(7):
        t0 = time.perf_counter()
        res = e.refine_sequence(frames, depths, K, init_t, opts, windows_per_call=wpc)
        times.append(time.perf_counter() - t0)
    med = sorted(times)[3]
    print(json.dumps({"path": f"tcsfm_refine_sequence: the window loop inside the library, {lanes} lane(s), {wpc} window(s) per call", "frames": T, "windows": T - 1,
                      "ms_total": round(med * 1e3, 2), "windows_per_s": round((T - 1) / med, 1),
                      "bit_identical_to_streamed": bool(torch.equal(res, ref.cpu()))}), flush=True)
    e.close()
# (d) with the reference's pose initialisation inside the loop: per window the coupled PoseNet loop (4 evaluations + 3 warps), then the
#     refinement -- tcsfm_odometry_sequence against the same work issued window by window from Python (frames already on the device)
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
import standins
from tightly_coupled_sfm_amd.posenet import PoseNetHIP
params = standins.posenet_params(0)
e = Engine(H, W, 2)
net = PoseNetHIP(e, 2, params)
fd, dd = frames.cuda(), depths.cuda(); Kd1 = torch.as_tensor(K[None]).cuda()
def per_window():
    outs, inits = [], []
    for w in range(T - 1):
        p0, _ = net.solve_pose_iteratively(4, fd[w][None], fd[w + 1][None, None], dd[w][None], dd[w + 1][None, None], Kd1)
        inits.append(p0)
        outs.append(e.refine_window(fd[w][None], fd[w + 1][None, None], dd[w][None], dd[w + 1][None, None], Kd1, p0, opts)[0])
    torch.cuda.synchronize()
    return torch.stack(outs), torch.stack(inits)
ref_o, ref_i = per_window()
ts = []
for rep in range(3):
    t0 = time.perf_counter(); per_window(); ts.append(time.perf_counter() - t0)
print(json.dumps({"path": "PoseNet loop (4 its) + refinement per window, one window after the other, device-resident frames (Python loop over the library calls)",
                  "windows_per_s": round((T - 1) / sorted(ts)[1], 1), "us_per_window": round(sorted(ts)[1] / (T - 1) * 1e6, 1)}), flush=True)
e.close()
for lanes, wpc in ((1, 1), (2, 1), (3, 1), (1, 8), (2, 8), (3, 8), (2, 16)):
    e = Engine(H, W, 2 * wpc, lanes=lanes)
    net = PoseNetHIP(e, 2 * wpc, params)
    net.odometry_sequence(frames[:40], depths[:40], K, opts, iterations=4, windows_per_call=wpc)
    ts = []
    for rep in range(5):
        t0 = time.perf_counter(); init_o, out_o = net.odometry_sequence(frames, depths, K, opts, iterations=4, windows_per_call=wpc); ts.append(time.perf_counter() - t0)
    print(json.dumps({"path": f"tcsfm_odometry_sequence: PoseNet loop (4 its) + refinement inside the library, frames streamed from pinned memory, {lanes} lane(s), {wpc} window(s) per call",
                      "windows_per_s": round((T - 1) / sorted(ts)[2], 1), "us_per_window": round(sorted(ts)[2] / (T - 1) * 1e6, 1),
                      "posenet_poses_max_rel_diff_to_per_window_calls": float((init_o - ref_i.cpu()).abs().max() / ref_i.abs().max()),
                      "refined_poses_max_rel_diff": float((out_o - ref_o.cpu()).abs().max() / ref_o.abs().max()),
                      "note": "the PoseNet's work split (hence its rounding, ~1e-6) depends on the number of images per call; the coupled loop has discrete decisions (warp validity), so with these RANDOM weights a few windows (3 of 199) amplify it to 1e-5..1e-2 -- all others agree to ~1e-6"}), flush=True)
    net.close(); e.close()
# (e) the reference's KITTI windows: 3 frames, target in the middle, sources = previous and next frame (S = 2, 4 directed pairs, min over
#     the sources, depth consistency), PoseNet loop + refinement
o2 = default_opts(n_iters=4, argmin=1, w_dc=0.15)
for lanes, wpc in ((1, 1), (2, 1), (3, 1), (1, 4), (2, 4), (2, 8)):
    e = Engine(H, W, 4 * wpc, lanes=lanes)
    net = PoseNetHIP(e, 4 * wpc, params)
    net.odometry_sequence(frames[:40], depths[:40], K, o2, sources=2, iterations=4, windows_per_call=wpc, target_pos=-1)
    ts = []
    for rep in range(5):
        t0 = time.perf_counter(); net.odometry_sequence(frames, depths, K, o2, sources=2, iterations=4, windows_per_call=wpc, target_pos=-1); ts.append(time.perf_counter() - t0)
    print(json.dumps({"path": f"tcsfm_odometry_sequence, KITTI windows (S = 2, target in the middle, min over sources, depth consistency), {lanes} lane(s), {wpc} window(s) per call",
                      "windows_per_s": round((T - 2) / sorted(ts)[2], 1), "us_per_window": round(sorted(ts)[2] / (T - 2) * 1e6, 1)}), flush=True)
    net.close(); e.close()
# the same windows from DEVICE-resident frames (no PCIe in the loop): what the lanes alone buy at B=1
dev_f, dev_d = frames.cuda(), depths.cuda()
Kd = torch.as_tensor(K[None]).cuda(); p0 = torch.as_tensor(init).cuda(); outd = torch.empty_like(p0)
for lanes in (1, 2, 3, 4):
    e = Engine(H, W, 2, lanes=lanes)
    tg = [dev_f[w][None] for w in range(T - 1)]; sc = [dev_f[w + 1][None, None] for w in range(T - 1)]
    dt = [dev_d[w][None] for w in range(T - 1)]; ds = [dev_d[w + 1][None, None] for w in range(T - 1)]
    pw, ow = list(p0.unbind(0)), list(outd.unbind(0))
    def sweep():
        for w in range(T - 1):
            e.refine_window_async(w % lanes, tg[w], sc[w], dt[w], ds[w], Kd, pw[w], ow[w], opts)
        for l in range(lanes):
            e.lane_synchronize(l)
    sweep(); torch.cuda.synchronize()
    times = []
    for rep in range(7):
        t0 = time.perf_counter(); sweep(); torch.cuda.synchronize(); times.append(time.perf_counter() - t0)
    med = sorted(times)[3]
    print(json.dumps({"path": f"device-resident frames, window form, {lanes} lane(s)", "windows_per_s": round((T - 1) / med, 1),
                      "us_per_window": round(med / (T - 1) * 1e6, 1), "equal_to_streamed": bool(torch.equal(outd, ref))}), flush=True)
    e.close()
