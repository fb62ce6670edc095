"""k_solve_front / k_solve_joint: phase timing of target 0's joint solve from in-kernel wall-clock stamps (TCSFM_DEBUG_STAMPS=2; 100 MHz ticks -> us):
   python scripts/diag/solve_front_stamps.py [H W S]"""
import ctypes as C, os, sys
import numpy as np, torch
os.environ["TCSFM_DEBUG_STAMPS"] = "2"
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ.setdefault("TCSFM_SET_ENV_DEFAULTS", "1")
from tightly_coupled_sfm_amd import _lib
from tightly_coupled_sfm_amd.engine import Engine, default_opts
import test_gpu_dense_reference as T
H, W, S = (int(x) for x in sys.argv[1:4]) if len(sys.argv) > 3 else (240, 320, 1)
B = int(os.environ.get("TCSFM_PROFILE_B", "1"))
w = T._window(B, S, H, W, seed=31)
t = {k: T._dev(v) for k, v in w.items()}
dt4, ds5 = t["depth_t"][:, None].contiguous(), t["depth_s"][:, :, None].contiguous()
e = Engine(H, W, 2 * S * B)
o = default_opts(n_iters=4, w_dc=0.15, prior_init=0.1, min_depth=0.06, max_depth=2.67, window_rule=_lib.WINDOW_REFERENCE)
acc = []
for i in range(120):
    e.refine_dense_window(t["tgt"], t["srcs"], dt4, ds5, t["K"], t["pose"], o, argmin=True)
    st = (C.c_longlong * 8)()
    e.lib.tcsfm_debug_stamps(e._h, st)
    if i >= 20:
        acc.append(np.array(st[:6], dtype=np.float64))
d = np.diff(np.stack(acc), axis=1) / 100.0
print(f"{H}x{W} S={S} B={B} phases us: record sum, cost / LM / assemble, Gauss-Jordan, step, retraction + constants + outputs")
print("mean", np.round(d.mean(0), 2), "total", round(d.sum(1).mean(), 2))
print("median", np.round(np.median(d, 0), 2))
