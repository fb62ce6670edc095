"""tcsfm_refine_sequence: PCIe-inclusive windows/s of a 200-frame 640x192 sequence against the number of lanes and ring slots"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import os as _os; _os.environ.setdefault("TCSFM_SET_ENV_DEFAULTS", "1")      # (a measurement script owns its process: HIP_FORCE_DEV_KERNARG / GPU_MAX_HW_QUEUES when absent)
from tightly_coupled_sfm_amd import synth
from tightly_coupled_sfm_amd.engine import Engine, default_opts
H, W, T = 192, 640, 200
seq = synth.make_sequence(T, H, W, seed=5)
frames = torch.as_tensor(seq["frames"]).pin_memory(); depths = torch.as_tensor(seq["depths"]).pin_memory()
K, init = seq["K"], torch.as_tensor(seq["init"])
opts = default_opts(n_iters=4)
for lanes in (2, 3, 4):
    for ring in (12, 16, 20, 24, 28, 32, 36, 48):
        e = Engine(H, W, 2, lanes=lanes)
        e.refine_sequence(frames[:40], depths[:40], K, init[:39], opts, ring=ring)
        ts = []
        for rep in range(5):
            t0 = time.perf_counter(); e.refine_sequence(frames, depths, K, init, opts, ring=ring); ts.append(time.perf_counter() - t0)
        print(json.dumps({"lanes": lanes, "ring": ring, "windows_per_s": round((T - 1) / sorted(ts)[2], 1), "min_max": [round((T-1)/max(ts)), round((T-1)/min(ts))]}), flush=True)
        e.close()
