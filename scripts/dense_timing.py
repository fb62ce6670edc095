"""dense mode timings (BASELINE config 5): per call and per k_dense_linearize launch (in-kernel bracket), B=1 and chip-full"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import os as _os; _os.environ.setdefault("TCSFM_SET_ENV_DEFAULTS", "1")      # (a measurement script owns its process: HIP_FORCE_DEV_KERNARG / GPU_MAX_HW_QUEUES when absent)
from tightly_coupled_sfm_amd import synth
from tightly_coupled_sfm_amd.engine import Engine, default_opts

BASIC = "basic" in sys.argv[1:]        # only the per-call / per-launch rows (A/B runs: scripts/experiments/dense_ab.sh)


def run(H, W, npairs, steps):
    b = synth.make_batch(2, H, W, seed0=0, both_directions=True)
    d = {k: torch.as_tensor(v).cuda().contiguous() for k, v in b.items()}
    if npairs > 2:
        d = {k: v.repeat((npairs // 2,) + (1,) * (v.dim() - 1)).contiguous() for k, v in d.items()}
    e = Engine(H, W, npairs)
    o = default_opts(n_iters=4, min_depth=0.03, max_depth=3.0)
    step = lambda: e.refine_dense(d["tgt"], d["src"], d["depth_t"], d["depth_s"], d["K"], d["pose_init"], o)
    for _ in range(max(3, steps // 10)): step()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(steps): step()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / steps
    e.profile_begin()
    for _ in range(min(steps, 50)): step()
    pr = e.profile_end()
    lin = pr["linearize_kernel"][0] / max(pr["linearize_kernel"][1], 1) * 1e3
    print(json.dumps({"tile": os.environ.get("TCSFM_DENSE_TILE", "32"), "HxW": f"{H}x{W}", "pairs": npairs, "us_per_call": round(dt * 1e6, 1),
                      "windows_per_s": round(npairs / 2 / dt, 1), "k_dense_linearize_us": round(lin, 2),
                      "GBps_alg36": round(36 * H * W * npairs / lin / 1e3, 1), "Gpx_per_s": round(H * W * npairs / lin / 1e3, 2)}), flush=True)

def run_lanes(H, W, lanes, steps, replay=False):
    """B=1 windows kept in flight on the handle's lanes (tcsfm_refine_dense_window_async): throughput of independent windows"""
    b = synth.make_batch(2, H, W, seed0=0, both_directions=True)
    d = {k: torch.as_tensor(v).cuda().contiguous() for k, v in b.items()}
    e = Engine(H, W, 2, lanes=lanes)
    e.use_own_stream()
    if replay:
        e.set_graph_replay(2)      # every lane's (repeated) call as one HIP graph: one host launch instead of ~13
    torch.cuda.synchronize()
    o = default_opts(n_iters=4, min_depth=0.03, max_depth=3.0)
    tgt, srcs, dt_, ds_ = d["tgt"][0:1].contiguous(), d["src"][0:1].contiguous()[None], d["depth_t"][0:1].contiguous(), d["depth_s"][0:1].contiguous()[None]
    K, p0 = d["K"][0:1].contiguous(), torch.stack([d["pose_init"][0], d["pose_init"][1]]).contiguous()
    po = [torch.empty_like(p0) for _ in range(lanes)]
    do = [torch.empty((2, 1, H, W), device="cuda") for _ in range(lanes)]
    def sweep(n):
        for k in range(n):
            l = k % lanes
            e.refine_dense_window_async(l, tgt, srcs, dt_, ds_, K, p0, po[l], do[l], o)
        for l in range(lanes):
            e.lane_synchronize(l)
    sweep(steps // 5)
    ts = []
    for _ in range(5):
        t0 = time.perf_counter(); sweep(steps); ts.append(time.perf_counter() - t0)
    t = sorted(ts)[2]
    same = all(torch.equal(po[0], x) for x in po) and all(torch.equal(do[0], x) for x in do)
    print(json.dumps({"HxW": f"{H}x{W}", "windows_per_call": 1, "calls_in_flight": lanes, "launches": "graph replay" if replay else "plain", "us_per_window": round(t / steps * 1e6, 1),
                      "windows_per_s": round(steps / t, 1), "lanes_agree": same}), flush=True)

for H, W in ((240, 320), (256, 448)):
    run(H, W, 2, 300)
    run(H, W, 64, 30)
    if BASIC:
        continue
    for lanes in (1, 2, 3):
        run_lanes(H, W, lanes, 600)
    for lanes in (2, 3):
        run_lanes(H, W, lanes, 600, replay=True)


def run_sequence(H, W, lanes, wpc, T=120):
    """the dense mode over a streamed sequence (tcsfm_refine_dense_sequence): frames from pinned host memory, depth maps back to the host"""
    seq = synth.make_sequence(T, H, W, seed=3)
    frames, depths = torch.as_tensor(seq["frames"]).pin_memory(), torch.as_tensor(seq["depths"]).pin_memory()
    init = torch.as_tensor(seq["init"])
    e = Engine(H, W, 2 * wpc, lanes=lanes)
    o = default_opts(n_iters=4, min_depth=0.03, max_depth=3.0)
    e.refine_dense_sequence(frames[:40], depths[:40], seq["K"], init[:39], o, windows_per_call=wpc)
    dout = torch.empty((T - 1, 2, 1, H, W), dtype=torch.float32, pin_memory=True)      # the caller's result buffer, reused (pinning 70 MB per call costs 10x the refinement)
    ts = []
    for _ in range(7):
        t0 = time.perf_counter(); e.refine_dense_sequence(frames, depths, seq["K"], init, o, windows_per_call=wpc, out_depths=dout); ts.append(time.perf_counter() - t0)
    t = sorted(ts)[3]
    print(json.dumps({"HxW": f"{H}x{W}", "path": "tcsfm_refine_dense_sequence (PCIe-inclusive, depth maps back to the host)", "lanes": lanes, "windows_per_call": wpc,
                      "us_per_window": round(t / (T - 1) * 1e6, 1), "windows_per_s": round((T - 1) / t, 1)}), flush=True)
    e.close()

for lanes, wpc in (() if BASIC else ((1, 1), (2, 1), (1, 8), (2, 8))):
    run_sequence(240, 320, lanes, wpc)
