"""Where the time of DepthOptimizer.optimize_window goes beside the engine call (bench.py `shim`): cProfile over 100 windows with the stand-in
networks, top entries by own time and by cumulative time.  usage: python scripts/diag/shim_profile.py [pose|dense]"""
import cProfile
import io
import os
import pstats
import sys

os.environ.setdefault("TCSFM_SET_ENV_DEFAULTS", "1")
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import standins
from tightly_coupled_sfm_amd.optimizer import DepthOptimizer

mode = sys.argv[1] if len(sys.argv) > 1 else "pose"
B, S, ITER, H, W = 1, 2, 3, 192, 640
w = standins.make_window(B, S, H, W)
cfg = {"minibatch": B, "device": "cuda", "min_depth": 0.06, "max_depth": 2.67, "iterations": ITER, "camera_height": 1.65, "flow_type": "none"}
opts = {"epochs": 5, "diff_img_argmin": True, "automasking": True, "mode": "scaled", "l_depth_consist": True, "l_depth_consist_weight": 0.15,
        "l_depth_init": True, "l_depth_init_weight": 0.1, "num_source_imgs": S, "avg_final_epochs": 5, "optimize_depth_pred": mode == "dense"}
pm, dm = standins.window_models(w, ITER, device="cuda")
opt = DepthOptimizer(opts, cfg, pm, dm, "09_02")
data = standins.loader_batch(w, device="cuda")
for _ in range(5):
    opt.optimize_window(0, data)
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for _ in range(100):
    opt.optimize_window(0, data)
torch.cuda.synchronize()
pr.disable()
for key in ("tottime", "cumulative"):
    s = io.StringIO()
    pstats.Stats(pr, stream=s).sort_stats(key).print_stats(38)
    print(s.getvalue()[:9000])
