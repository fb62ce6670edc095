"""Wall time of DepthOptimizer.optimize_window (the drop-in for the reference's optimiser) with stand-in networks."""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import standins
import os as _os; _os.environ.setdefault("TCSFM_SET_ENV_DEFAULTS", "1")      # (a measurement script owns its process: HIP_FORCE_DEV_KERNARG / GPU_MAX_HW_QUEUES when absent)
from tightly_coupled_sfm_amd.optimizer import DepthOptimizer
B, S, H, W, ITER = 1, 2, 192, 640, 3
w = standins.make_window(B, S, H, W)
opts = {"epochs": 5, "optimize_depth_pred": False, "diff_img_argmin": True, "automasking": True, "mode": "scaled", "l_depth_consist": True,
        "l_depth_consist_weight": 0.15, "num_source_imgs": S, "avg_final_epochs": 5}
cfg = {"minibatch": B, "device": "cuda", "min_depth": 0.06, "max_depth": 2.67, "iterations": ITER, "camera_height": 1.65, "flow_type": "none"}
pm, dm = standins.window_models(w, ITER, device="cuda")
opt = DepthOptimizer(opts, cfg, pm, dm, "09_02")
data = standins.loader_batch(w, device="cuda")
for _ in range(5): opt.optimize_window(0, data)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(30): opt.optimize_window(0, data)
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 30
print(f"optimize_window (B={B}, S={S}, {W}x{H}, PoseNet iterations {ITER}, 4 GN its, argmin): {dt * 1e3:.2f} ms per window")
import cProfile, pstats
pr = cProfile.Profile(); pr.enable()
for _ in range(10): opt.optimize_window(0, data)
torch.cuda.synchronize(); pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(18)
