#!/usr/bin/env python3
"""dense window mode on the reference's loss (window_rule REFERENCE) against the library's joint / per-pair dense modes: time per window
(one call in flight, device pointers), per size and source count -> one JSON line each (profiles/r04_dense_ref_timing.jsonl)"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
from tightly_coupled_sfm_amd import _lib
from tightly_coupled_sfm_amd.engine import Engine, default_opts
import test_gpu_dense_reference as T

for H, W, S, mind, maxd in ((240, 320, 1, 0.03, 3.0), (256, 448, 1, 0.03, 3.0), (192, 640, 2, 0.06, 2.67)):
    w = T._window(1, S, H, W, seed=31)
    t = {k: T._dev(v) for k, v in w.items()}
    dt4, ds5 = t["depth_t"][:, None].contiguous(), t["depth_s"][:, :, None].contiguous()
    e = Engine(H, W, 2 * S)
    for name, o in (("reference loss (forward + inverse + depth consistency + SSIM prior)", default_opts(n_iters=4, w_dc=0.15, prior_init=0.1, min_depth=mind, max_depth=maxd, window_rule=_lib.WINDOW_REFERENCE)),
                    ("reference loss, QUARTER-resolution unknown (the reference's parametrisation, optimizer.py:194-198)", default_opts(n_iters=4, w_dc=0.15, prior_init=0.1, min_depth=mind, max_depth=maxd, window_rule=_lib.WINDOW_REFERENCE, depth_param=_lib.DEPTH_QUARTER)),
                    ("reference loss, the SOURCE depth maps unknowns too (free_source_depths; every leaf of optimize_depth_pred moves)", default_opts(n_iters=4, w_dc=0.15, prior_init=0.1, min_depth=mind, max_depth=maxd, window_rule=_lib.WINDOW_REFERENCE, free_source_depths=1)),
                    ("reference loss, the reference's complete leaf set: QUARTER-resolution maps of target AND sources (depth_param QUARTER + free_source_depths)", default_opts(n_iters=4, w_dc=0.15, prior_init=0.1, min_depth=mind, max_depth=maxd, window_rule=_lib.WINDOW_REFERENCE, depth_param=_lib.DEPTH_QUARTER, free_source_depths=1)),
                    ("reference loss, forward + inverse only", default_opts(n_iters=4, w_dc=0.0, prior_init=0.0, min_depth=mind, max_depth=maxd, window_rule=_lib.WINDOW_REFERENCE)),
                    ("library joint / pair dense mode (own weights, Tikhonov prior)", default_opts(n_iters=4, min_depth=mind, max_depth=maxd))):
        step = lambda: e.refine_dense_window(t["tgt"], t["srcs"], dt4, ds5, t["K"], t["pose"], o, argmin=True)
        for _ in range(20): step()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(200): step()
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 200
        e.profile_begin()
        for _ in range(30): step()
        pr = e.profile_end()
        lin = pr["linearize_kernel"][0] / max(pr["linearize_kernel"][1], 1) * 1e3
        print(json.dumps({"HxW": f"{H}x{W}", "S": S, "mode": name, "us_per_window": round(dt * 1e6, 1), "windows_per_s": round(1 / dt, 1),
                          "joint_kernel_us": round(lin, 2)}), flush=True)
