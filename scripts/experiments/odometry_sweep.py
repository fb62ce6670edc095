"""tcsfm_odometry_sequence (PoseNet loop 4 iterations + refinement, KITTI windows S=2) over lanes x windows per call: one JSON line each.
ON THE GPU BOX: python scripts/experiments/odometry_sweep.py [lanes:wpc ...]"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import standins
import os as _os; _os.environ.setdefault("TCSFM_SET_ENV_DEFAULTS", "1")      # (a measurement script owns its process: HIP_FORCE_DEV_KERNARG / GPU_MAX_HW_QUEUES when absent)
from tightly_coupled_sfm_amd import synth
from tightly_coupled_sfm_amd.engine import Engine, default_opts
from tightly_coupled_sfm_amd.posenet import PoseNetHIP
H, W, T, S = 192, 640, 200, 2
seq = synth.make_sequence(T, H, W, seed=5)
frames = torch.as_tensor(seq["frames"]).pin_memory(); depths = torch.as_tensor(seq["depths"]).pin_memory()
K = seq["K"]
o2 = default_opts(n_iters=4, argmin=1, w_dc=0.15)
params = standins.posenet_params(0)
cfgs = [tuple(int(x) for x in a.split(":")) for a in sys.argv[1:]] or [(1, 8), (2, 8), (3, 8), (4, 8), (2, 4), (4, 4), (2, 16), (3, 16)]
for lanes, wpc in cfgs:
    e = Engine(H, W, 2 * S * wpc, lanes=lanes)
    net = PoseNetHIP(e, 2 * S * wpc, params)
    kw = dict(sources=S, iterations=4, windows_per_call=wpc, target_pos=-1)
    net.odometry_sequence(frames[:40], depths[:40], K, o2, **kw)
    ts = []
    for _ in range(5):
        t0 = time.perf_counter(); net.odometry_sequence(frames, depths, K, o2, **kw); ts.append(time.perf_counter() - t0)
    print(json.dumps({"lanes": lanes, "windows_per_call": wpc, "windows_per_s": round((T - S) / sorted(ts)[2], 1)}), flush=True)
    net.close(); e.close()
