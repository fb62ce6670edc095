"""Sequence throughput with the frame-level pack cache (PCIe-inclusive, 200 frames of 640x192 from pinned memory):
S=1 refine sequence, S=2 KITTI windows (refine only, and with the PoseNet loop)   -> one JSON line per configuration"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
import standins
import os as _os; _os.environ.setdefault("TCSFM_SET_ENV_DEFAULTS", "1")      # (a measurement script owns its process: HIP_FORCE_DEV_KERNARG / GPU_MAX_HW_QUEUES when absent)
from tightly_coupled_sfm_amd import synth
from tightly_coupled_sfm_amd.engine import Engine, default_opts
from tightly_coupled_sfm_amd.posenet import PoseNetHIP
H, W, T = 192, 640, 200
seq = synth.make_sequence(T, H, W, seed=5)
frames = torch.as_tensor(seq["frames"]).pin_memory(); depths = torch.as_tensor(seq["depths"]).pin_memory()
K, init = seq["K"], torch.as_tensor(seq["init"])
def med(f, n=7):
    ts = []
    for _ in range(n):
        t0 = time.perf_counter(); f(); ts.append(time.perf_counter() - t0)
    return sorted(ts)[n // 2]
opts = default_opts(n_iters=4)
for lanes, wpc in ((1, 8), (2, 8), (2, 16)):
    e = Engine(H, W, 2 * wpc, lanes=lanes)
    e.refine_sequence(frames[:60], depths[:60], K, init[:59], opts, windows_per_call=wpc)
    t = med(lambda: e.refine_sequence(frames, depths, K, init, opts, windows_per_call=wpc))
    print(json.dumps({"path": f"tcsfm_refine_sequence S=1, {lanes} lane(s), {wpc} windows per call", "windows_per_s": round((T - 1) / t, 1)}), flush=True)
    e.close()
o2 = default_opts(n_iters=4, argmin=1, w_dc=0.15)
step = seq["init"][:, 0]
init2 = torch.as_tensor(np.stack([np.stack([-step[w], step[w + 1], step[w], -step[w + 1]]) for w in range(T - 2)]).astype(np.float32))
for lanes, wpc in ((1, 8), (2, 8)):
    e = Engine(H, W, 4 * wpc, lanes=lanes)
    e.refine_sequence(frames[:40], depths[:40], K, init2[:38], o2, sources=2, windows_per_call=wpc, target_pos=-1)
    t = med(lambda: e.refine_sequence(frames, depths, K, init2, o2, sources=2, windows_per_call=wpc, target_pos=-1))
    print(json.dumps({"path": f"tcsfm_refine_sequence KITTI windows (S=2, target in the middle, argmin, w_dc), {lanes} lane(s), {wpc} windows per call",
                      "windows_per_s": round((T - 2) / t, 1)}), flush=True)
    e.close()
params = standins.posenet_params(0)
for lanes, wpc, S, o in ((2, 8, 1, opts), (2, 8, 2, o2)):
    e = Engine(H, W, 2 * S * wpc, lanes=lanes)
    net = PoseNetHIP(e, 2 * S * wpc, params)
    kw = dict(sources=S, iterations=4, windows_per_call=wpc, target_pos=-1 if S == 2 else 0)
    net.odometry_sequence(frames[:40], depths[:40], K, o, **kw)
    t = med(lambda: net.odometry_sequence(frames, depths, K, o, **kw), n=5)
    print(json.dumps({"path": f"tcsfm_odometry_sequence S={S} (PoseNet loop 4 its + refinement), {lanes} lanes, {wpc} windows per call",
                      "windows_per_s": round((T - S) / t, 1)}), flush=True)
    net.close(); e.close()
