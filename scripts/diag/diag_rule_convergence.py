"""Convergence of the two window rules on the float64 oracle (CPU): distance of the k-iteration forward poses to the same rule's
300-iteration limit, translation (relative) and rotation (deg), and the gradient norm at the iterate."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle.oracle import Oracle, default_opts
import os as _os; _os.environ.setdefault("TCSFM_SET_ENV_DEFAULTS", "1")      # (a measurement script owns its process: HIP_FORCE_DEV_KERNARG / GPU_MAX_HW_QUEUES when absent)
from tightly_coupled_sfm_amd import synth
H, W, S, B = 96, 320, 2, 1
orc = Oracle("f64")
pose = np.array([0.002, -0.001, 0.033, 0.001, -0.003, 0.001])
pairs = [synth.make_pair(H, W, seed=500, pose_gt=pose * (1 if s == 0 else -1)) for s in range(S)]
tgt = pairs[0]["tgt"][None]; srcs = np.stack([p["src"] for p in pairs])[:, None]
dt = pairs[0]["depth_t"][None, None]; ds = np.stack([p["depth_s"] for p in pairs])[:, None, None]
K = pairs[0]["K"][None]
fwd = np.stack([synth.perturb_pose(p["pose_gt"], 900 + s) for s, p in enumerate(pairs)])
p0 = np.concatenate([fwd, np.stack([synth.invert_pose(x) for x in fwd])])
for rule in (0, 1):
    lim = orc.refine_window(tgt, srcs, dt, ds, K, p0, default_opts(n_iters=300, w_dc=0.15), argmin=True, rule=rule)[0]
    for k in (0, 1, 2, 4, 8, 16, 40, 100):
        pk = p0 if k == 0 else orc.refine_window(tgt, srcs, dt, ds, K, p0, default_opts(n_iters=k, w_dc=0.15), argmin=True, rule=rule)[0]
        L = orc.linearize_window(tgt, srcs, dt, ds, K, pk, default_opts(n_iters=1, w_dc=0.15), argmin=True, rule=rule)
        et = np.linalg.norm(pk[:S, :3] - lim[:S, :3], axis=1) / np.linalg.norm(lim[:S, :3], axis=1)
        er = np.degrees(np.linalg.norm(pk[:S, 3:] - lim[:S, 3:], axis=1))
        print(f"rule {rule} k={k:3d} trans {et.round(4)} rot_deg {er.round(4)} cost {L['cost'].sum():.6f} |g| fwd {np.linalg.norm(L['g'][:S], axis=1).round(6)}")
