"""k_solve phase timing from the in-kernel wall-clock stamps (TCSFM_DEBUG_STAMPS=1; 100 MHz ticks -> us)."""
import ctypes as C, os, sys
import numpy as np, torch
os.environ["TCSFM_DEBUG_STAMPS"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import os as _os; _os.environ.setdefault("TCSFM_SET_ENV_DEFAULTS", "1")      # (a measurement script owns its process: HIP_FORCE_DEV_KERNARG / GPU_MAX_HW_QUEUES when absent)
from tightly_coupled_sfm_amd import synth
from tightly_coupled_sfm_amd.engine import Engine, default_opts
H, W = 192, 640
b = synth.make_batch(2, H, W, seed0=0, both_directions=True)
dev = {k: torch.as_tensor(v).cuda().contiguous() for k, v in b.items()}
e = Engine(H, W, 2)
o = default_opts(n_iters=4)
out = torch.empty_like(dev["pose_init"])
acc = []
for i in range(200):
    e.refine_into(dev["tgt"], dev["src"], dev["depth_t"], dev["depth_s"], dev["K"], dev["pose_init"], out, o)
    st = (C.c_longlong * 8)()
    e.lib.tcsfm_debug_stamps(e._h, st)
    if i >= 20:
        acc.append(np.array(st[:7], dtype=np.float64))
a = np.diff(np.stack(acc), axis=1) / 100.0
print("phases us: reduce, assemble, gauss-jordan, retract, write_const, pose_out")
print("mean", np.round(a.mean(0), 3), "total", round(a.sum(1).mean(), 3))
print("median", np.round(np.median(a, 0), 3))
