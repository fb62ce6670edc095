#!/usr/bin/env python3
"""workload for rocprofv3 --kernel-trace --stats: the dense window mode on the reference's loss, 240x320 S=1 and 192x640 S=2, 4 GN iterations; a fourth argument `quarter` selects the quarter-resolution unknown; TCSFM_PROFILE_B = windows per call (default 1; 6 = the reference's minibatch)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
import os as _os; _os.environ.setdefault("TCSFM_SET_ENV_DEFAULTS", "1")      # (a measurement script owns its process: HIP_FORCE_DEV_KERNARG / GPU_MAX_HW_QUEUES when absent)
from tightly_coupled_sfm_amd import _lib
from tightly_coupled_sfm_amd.engine import Engine, default_opts
import test_gpu_dense_reference as T
H, W, S = (int(x) for x in sys.argv[1:4]) if len(sys.argv) > 3 else (192, 640, 2)
QUARTER = len(sys.argv) > 4 and sys.argv[4] in ("quarter", "quarter_free")
FREE = len(sys.argv) > 4 and sys.argv[4] in ("free", "quarter_free")      # the source maps unknowns too (free_source_depths)
B = int(os.environ.get("TCSFM_PROFILE_B", "1"))
w = T._window(B, S, H, W, seed=31)
t = {k: T._dev(v) for k, v in w.items()}
dt4, ds5 = t["depth_t"][:, None].contiguous(), t["depth_s"][:, :, None].contiguous()
e = Engine(H, W, 2 * S * B)
o = default_opts(n_iters=4, w_dc=0.15, prior_init=0.1, min_depth=0.06, max_depth=2.67, window_rule=_lib.WINDOW_REFERENCE,
                 depth_param=_lib.DEPTH_QUARTER if QUARTER else _lib.DEPTH_FULL, free_source_depths=1 if FREE else 0)
for _ in range(100):
    e.refine_dense_window(t["tgt"], t["srcs"], dt4, ds5, t["K"], t["pose"], o, argmin=True)
torch.cuda.synchronize()
