import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import standins
from oracle.oracle import Oracle, default_opts as oopts
import os as _os; _os.environ.setdefault("TCSFM_SET_ENV_DEFAULTS", "1")      # (a measurement script owns its process: HIP_FORCE_DEV_KERNARG / GPU_MAX_HW_QUEUES when absent)
from tightly_coupled_sfm_amd.engine import Engine, default_opts
B, S, H, W = 2, 2, 96, 320
w = standins.make_window(B, S, H, W, seed0=90)
o64 = Oracle("f64")
w["depth_t"] = o64.disp_to_depth(w["disp_t"], 0.06, 2.67)[1].astype(np.float32)
w["depth_s"] = o64.disp_to_depth(w["disp_s"], 0.06, 2.67)[1].astype(np.float32)
t = lambda a: torch.as_tensor(np.ascontiguousarray(a, dtype=np.float32)).cuda()
e = Engine(H, W, 2 * S * B)
for argmin in (False, True):
    for iters in (1, 2, 4):
        pose, _, st = e.refine_window(t(w["target"]), t(w["sources"]), t(w["depth_t"]), t(w["depth_s"]), t(w["K"]), t(w["first"]), default_opts(n_iters=iters), stats=True, argmin=argmin)
        rp, _, rst = o64.refine_window(w["target"], w["sources"], w["depth_t"][:, 0], w["depth_s"][:, :, 0], w["K"], w["first"], oopts(n_iters=iters), argmin=argmin)
        pose = pose.cpu().numpy().astype(np.float64); st = st.cpu().numpy()
        print("argmin", argmin, "iters", iters)
        for n in range(2 * S * B):
            et = np.linalg.norm(pose[n, :3] - rp[n, :3]) / np.linalg.norm(rp[n, :3]); er = np.linalg.norm(pose[n, 3:] - rp[n, 3:]) / np.linalg.norm(rp[n, 3:])
            print(f"  pair {n}: et {et:.2e} er {er:.2e} nmask gpu {st[n, :iters, 2]} orc {rst[n, :iters, 2]} cost {st[n,:iters,0]} {rst[n,:iters,0]}")
