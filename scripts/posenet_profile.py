"""one PoseNet configuration for rocprofv3 --kernel-trace: python scripts/posenet_profile.py [N images] [calls]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
import standins
import os as _os; _os.environ.setdefault("TCSFM_SET_ENV_DEFAULTS", "1")      # (a measurement script owns its process: HIP_FORCE_DEV_KERNARG / GPU_MAX_HW_QUEUES when absent)
from tightly_coupled_sfm_amd.engine import Engine
from tightly_coupled_sfm_amd.posenet import PoseNetHIP
N = int(sys.argv[1]) if len(sys.argv) > 1 else 2
calls = int(sys.argv[2]) if len(sys.argv) > 2 else 50
H, W = 192, 640
x = torch.rand((N, 6, H, W), device="cuda")
net = PoseNetHIP(Engine(H, W, N), N, standins.posenet_params(0))
for _ in range(calls):
    out = net(x)
torch.cuda.synchronize()
print("done", float(out.abs().sum()))
