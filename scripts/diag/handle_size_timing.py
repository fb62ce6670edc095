"""Does a B=1 call cost more on a handle sized for many pairs?  (dense modes, 192x640 S=2; one call in flight)
usage: python scripts/diag/handle_size_timing.py"""
import os, sys, time
os.environ.setdefault("TCSFM_SET_ENV_DEFAULTS", "1")
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from tightly_coupled_sfm_amd import _lib
from tightly_coupled_sfm_amd.engine import Engine, default_opts
import test_gpu_dense_reference as T

H, W, S, B = 192, 640, 2, 1
w = T._window(B, S, H, W, seed=31)
t = {k: T._dev(v) for k, v in w.items()}
dt4, ds5 = t["depth_t"][:, None].contiguous(), t["depth_s"][:, :, None].contiguous()
kw = dict(n_iters=4, min_depth=0.06, max_depth=2.67)
modes = (("lib joint dense (PAIR rule)", default_opts(**kw)),
         ("reference loss", default_opts(window_rule=_lib.WINDOW_REFERENCE, w_dc=0.15, prior_init=0.1, **kw)),
         ("pose window (PAIR)", None))
for mp in (4, 32, 128):
    e = Engine(H, W, mp)
    for name, o in modes:
        if o is None:
            oo = default_opts(**kw)
            step = lambda: e.refine_window(t["tgt"], t["srcs"], dt4, ds5, t["K"], t["pose"], oo, argmin=True)
        else:
            step = lambda: e.refine_dense_window(t["tgt"], t["srcs"], dt4, ds5, t["K"], t["pose"], o, argmin=True)
        for _ in range(10): step()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(100): step()
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 100
        print(f"max_pairs {mp:4d}  {name:32s} {dt * 1e6:8.1f} us per call", flush=True)
    e.close()
