"""one tcsfm_odometry_sequence configuration for rocprofv3 --kernel-trace --stats: python scripts/odometry_profile.py [S] [windows per call] [lanes] [frames]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch, standins
import os as _os; _os.environ.setdefault("TCSFM_SET_ENV_DEFAULTS", "1")      # (a measurement script owns its process: HIP_FORCE_DEV_KERNARG / GPU_MAX_HW_QUEUES when absent)
from tightly_coupled_sfm_amd import synth
from tightly_coupled_sfm_amd.engine import Engine, default_opts
from tightly_coupled_sfm_amd.posenet import PoseNetHIP
S = int(sys.argv[1]) if len(sys.argv) > 1 else 1
wpc = int(sys.argv[2]) if len(sys.argv) > 2 else 8
lanes = int(sys.argv[3]) if len(sys.argv) > 3 else 2
T = int(sys.argv[4]) if len(sys.argv) > 4 else 100
H, W = 192, 640
seq = synth.make_sequence(T, H, W, seed=5)
frames, depths = torch.as_tensor(seq["frames"]).pin_memory(), torch.as_tensor(seq["depths"]).pin_memory()
e = Engine(H, W, 2 * S * wpc, lanes=lanes)
net = PoseNetHIP(e, 2 * S * wpc, standins.posenet_params(0))
o = default_opts(n_iters=4, argmin=1, w_dc=0.15 if S > 1 else 0.0)
for _ in range(3):
    init, out = net.odometry_sequence(frames, depths, seq["K"], o, sources=S, iterations=4, windows_per_call=wpc, target_pos=-1)
print("done", float(out.abs().sum()))
