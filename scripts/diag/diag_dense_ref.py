#!/usr/bin/env python3
"""diagnostic: dense REFERENCE mode, engine vs float64 oracle after k Gauss-Newton iterations (poses, depth map), term by term"""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "tests"))
import numpy as np, torch
from oracle.oracle import Oracle, default_opts as oo_
import os as _os; _os.environ.setdefault("TCSFM_SET_ENV_DEFAULTS", "1")      # (a measurement script owns its process: HIP_FORCE_DEV_KERNARG / GPU_MAX_HW_QUEUES when absent)
from tightly_coupled_sfm_amd.engine import Engine, default_opts
from tightly_coupled_sfm_amd import _lib
import test_gpu_dense_reference as T

H, W, S = (int(x) for x in sys.argv[1:4]) if len(sys.argv) > 3 else (96, 160, 1)
orc = Oracle("f64")
w = T._window(1, S, H, W, seed=31)
N = 2 * S
f32 = lambda a: np.asarray(a, dtype=np.float32).astype(np.float64)
t = {k: T._dev(v) for k, v in w.items()}
dt4, ds5 = t["depth_t"][:, None].contiguous(), t["depth_s"][:, :, None].contiguous()
e = Engine(H, W, N)
for cfg in (dict(w_dc=0.0, prior_init=0.0), dict(w_dc=0.15, prior_init=0.0), dict(w_dc=0.0, prior_init=0.1), dict(w_dc=0.15, prior_init=0.1)):
    for n_it in (1, 2):
        o = default_opts(n_iters=n_it, min_depth=0.03, max_depth=3.0, window_rule=_lib.WINDOW_REFERENCE, lambda_depth=1.0, **cfg)
        pose, depth, st = e.refine_dense_window(t["tgt"], t["srcs"], dt4, ds5, t["K"], t["pose"], o, stats=True, argmin=True)
        pose = pose.cpu().numpy().astype(np.float64); depth = depth.cpu().numpy().astype(np.float64)
        po, do, so = orc.refine_dense_ref(f32(w["tgt"]), f32(w["srcs"]), f32(w["depth_t"]), f32(w["depth_s"]), f32(w["K"]), f32(w["pose"]),
                                          oo_(n_iters=n_it, w_dc=cfg["w_dc"]), argmin=True, w_init=cfg["prior_init"], lambda_depth=1.0, min_depth=0.03, max_depth=3.0)
        ep = [np.linalg.norm(pose[m] - po[m]) / np.linalg.norm(po[m]) for m in range(N)]
        rel = np.abs(depth[0, 0] / do[0] - 1)
        mv = np.abs(do[0] / f32(w["depth_t"])[0] - 1)
        print(cfg, "iters", n_it, "pose rel err", ["%.1e" % x for x in ep], "depth rel err max %.2e median %.2e (moved max %.2e) at" % (rel.max(), np.median(rel), mv.max()),
              np.unravel_index(rel.argmax(), rel.shape), "stats fwd cost", st[0, :n_it, 0].cpu().numpy(), "oracle L", so[:, 0], so[:, 1] + so[:, 4])
