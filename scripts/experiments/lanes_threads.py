"""B=1 windows from device-resident frames: lanes driven by ONE host thread vs one host thread PER lane (the refine call spends
~50 us enqueueing 9 kernel launches; ctypes releases the GIL for that time, so threads overlap the enqueue cost)"""
import json, os, sys, time, threading
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import os as _os; _os.environ.setdefault("TCSFM_SET_ENV_DEFAULTS", "1")      # (a measurement script owns its process: HIP_FORCE_DEV_KERNARG / GPU_MAX_HW_QUEUES when absent)
from tightly_coupled_sfm_amd import synth
from tightly_coupled_sfm_amd.engine import Engine, default_opts

H, W, NW = 192, 640, 400
b = synth.make_batch(2, H, W, seed0=0, both_directions=True)
d = {k: torch.as_tensor(v).cuda().contiguous() for k, v in b.items()}
tgt, src = d["tgt"][0:1], d["src"][0:1][None]
dt, ds = d["depth_t"][0:1], d["depth_s"][0:1][None]
K = d["K"][0:1]
p0 = torch.stack([d["pose_init"][0], d["pose_init"][1]])
opts = default_opts(n_iters=4)
for lanes in (1, 2, 3, 4):
    for threaded in (False, True):
        if threaded and lanes == 1:
            continue
        e = Engine(H, W, 2, lanes=lanes)
        outs = [torch.empty_like(p0) for _ in range(lanes)]
        def drive(l, n):
            for _ in range(n):
                e.refine_window_async(l, tgt, src, dt, ds, K, p0, outs[l], opts)
        def sweep():
            if threaded:
                th = [threading.Thread(target=drive, args=(l, NW // lanes)) for l in range(lanes)]
                for t in th: t.start()
                for t in th: t.join()
            else:
                for w in range(NW):
                    e.refine_window_async(w % lanes, tgt, src, dt, ds, K, p0, outs[w % lanes], opts)
            for l in range(lanes):
                e.lane_synchronize(l)
        sweep(); torch.cuda.synchronize()
        times = []
        for _ in range(7):
            t0 = time.perf_counter(); sweep(); torch.cuda.synchronize(); times.append(time.perf_counter() - t0)
        med = sorted(times)[3]
        n_done = (NW // lanes) * lanes if threaded else NW
        ok = all(torch.equal(o, outs[0]) for o in outs)
        print(json.dumps({"lanes": lanes, "host_threads": lanes if threaded else 1, "windows_per_s": round(n_done / med, 1),
                          "us_per_window": round(med / n_done * 1e6, 1), "lanes_agree": ok}), flush=True)
