import os, sys
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import numpy as np, torch
import standins
from conftest import load_golden
from oracle.oracle import Oracle
import os as _os; _os.environ.setdefault("TCSFM_SET_ENV_DEFAULTS", "1")      # (a measurement script owns its process: HIP_FORCE_DEV_KERNARG / GPU_MAX_HW_QUEUES when absent)
from tightly_coupled_sfm_amd.engine import Engine
from tightly_coupled_sfm_amd.posenet import PoseNetHIP
g = load_golden("posenet")
B, S, H, W = 2, 2, 48, 160
w = standins.make_window(B, S, H, W, seed0=90)
o64 = Oracle("f64")
w["depth_t"] = o64.disp_to_depth(w["disp_t"], 0.06, 2.67)[1].astype(np.float32)
w["depth_s"] = o64.disp_to_depth(w["disp_s"], 0.06, 2.67)[1].astype(np.float32)
t = lambda a: torch.as_tensor(np.ascontiguousarray(a, dtype=np.float32)).cuda()
e = Engine(H, W, 8)
net = PoseNetHIP(e, 8, standins.posenet_params(0))
poses, stacked = net.solve_pose_iteratively(4, t(w["target"]), t(w["sources"]), t(w["depth_t"]), t(w["depth_s"]), t(w["K"]))
st = stacked.cpu().numpy(); ref = g["loop_stacked"]
for it in range(4):
    print(it, ["%.1e" % (np.abs(st[n, it] - ref[n, it]).max() / np.abs(ref[:, it]).max()) for n in range(8)])
# the same loop with the torch twin + library warps
twin = standins.PoseNetTwin(standins.posenet_params(0)).cuda().eval()
tg, sr = t(w["target"]), t(w["sources"])
T = tg.repeat(S, 1, 1, 1); Sx = sr.reshape(S * B, 3, H, W)
Dt = t(w["depth_t"]).repeat(S, 1, 1, 1); Ds = t(w["depth_s"]).reshape(S * B, 1, H, W)
tgt = torch.cat([T, Sx]); src = torch.cat([Sx, T]); dt = torch.cat([Dt, Ds]); ds = torch.cat([Ds, Dt]); K = t(w["K"]).repeat(2 * S, 1, 1)
with torch.no_grad():
    full = twin(torch.cat([tgt, src], 1))
    print("twin it0", ["%.1e" % (np.abs(full.cpu().numpy()[n] - ref[n, 0]).max() / np.abs(ref[:, 0]).max()) for n in range(8)])
    for it in range(1, 4):
        new = e.posenet_input(tgt, src, dt, ds, full.contiguous(), K)
        full = full + twin(new)
        print("twin+libwarp it", it, ["%.1e" % (np.abs(full.cpu().numpy()[n] - ref[n, it]).max() / np.abs(ref[:, it]).max()) for n in range(8)])
        # net on the same input
        mine = net(new)
        print("   hip net vs twin on this input:", float((mine - twin(new)).abs().max() / twin(new).abs().max()))
