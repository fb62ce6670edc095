#!/usr/bin/env python3
"""A/B helper (ab_modes.sh style): the per-pair dense mode (k_dense_linearize) with whatever library is in place -- one call at a time and a
64-window call at 320x240 and 448x256, plus a hash of refined poses / depth maps of fixed inputs."""
import hashlib, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import os as _os; _os.environ.setdefault("TCSFM_SET_ENV_DEFAULTS", "1")      # (a measurement script owns its process: HIP_FORCE_DEV_KERNARG / GPU_MAX_HW_QUEUES when absent)
from tightly_coupled_sfm_amd import synth
from tightly_coupled_sfm_amd.engine import Engine, default_opts
out = {}
h = hashlib.sha256()
opts = default_opts(n_iters=4, min_depth=0.03, max_depth=3.0)
for (H, W) in ((240, 320), (256, 448)):
    for npairs, steps, warm in ((2, 300, 30), (128, 20, 3)):
        b = synth.make_batch(npairs, H, W, seed0=0, both_directions=True)
        d = {k: torch.as_tensor(v).cuda().contiguous() for k, v in b.items()}
        e = Engine(H, W, npairs)
        fn = lambda: e.refine_dense(d["tgt"], d["src"], d["depth_t"], d["depth_s"], d["K"], d["pose_init"], opts)
        r = fn(); torch.cuda.synchronize()
        if npairs == 2:
            for t in r:
                if torch.is_tensor(t): h.update(t.detach().cpu().numpy().tobytes())
        for _ in range(warm): fn()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(steps): fn()
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / steps
        out[f"dense_{H}x{W}_pairs{npairs}_us"] = round(dt * 1e6, 1)
        e.close()
out["sha"] = h.hexdigest()[:12]
print(json.dumps(out))
