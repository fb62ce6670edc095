"""one streamed sequence configuration for rocprofv3: python scripts/seq_profile.py [S] [lanes] [windows_per_call]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import os as _os; _os.environ.setdefault("TCSFM_SET_ENV_DEFAULTS", "1")      # (a measurement script owns its process: HIP_FORCE_DEV_KERNARG / GPU_MAX_HW_QUEUES when absent)
from tightly_coupled_sfm_amd import synth
from tightly_coupled_sfm_amd.engine import Engine, default_opts
S = int(sys.argv[1]) if len(sys.argv) > 1 else 1
lanes = int(sys.argv[2]) if len(sys.argv) > 2 else 1
wpc = int(sys.argv[3]) if len(sys.argv) > 3 else 8
H, W, T = 192, 640, 200
seq = synth.make_sequence(T, H, W, seed=5)
frames = torch.as_tensor(seq["frames"]).pin_memory(); depths = torch.as_tensor(seq["depths"]).pin_memory()
step = seq["init"][:, 0]
if S == 1:
    init, o, tp = torch.as_tensor(seq["init"]), default_opts(n_iters=4), 0
else:
    init = torch.as_tensor(np.stack([np.stack([-step[w], step[w + 1], step[w], -step[w + 1]]) for w in range(T - 2)]).astype(np.float32))
    o, tp = default_opts(n_iters=4, argmin=1, w_dc=0.15), -1
e = Engine(H, W, 2 * S * wpc, lanes=lanes)
for _ in range(5):
    out = e.refine_sequence(frames, depths, seq["K"], init, o, sources=S, windows_per_call=wpc, target_pos=tp)
print("done", float(out.abs().sum()))
