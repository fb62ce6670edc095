"""diagnostic: depth-consistency part of the normal equations, engine vs oracle, pair 7 of the 96x320 test window"""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import parity_util as PU
from oracle.oracle import Oracle, default_opts as oopts
import os as _os; _os.environ.setdefault("TCSFM_SET_ENV_DEFAULTS", "1")      # (a measurement script owns its process: HIP_FORCE_DEV_KERNARG / GPU_MAX_HW_QUEUES when absent)
from tightly_coupled_sfm_amd.engine import Engine, default_opts
from test_gpu_window_rule import _window, _t
np.set_printoptions(linewidth=200, precision=3)
orc = Oracle("f64")
B, S, H, W = 2, 2, 96, 320
w = _window(B, S, H, W)
views = PU.window_pair_views(w)
for n in (5, 7):
    t, sr, dt, ds, K = views[n]
    pose = w["first"][n]
    e = Engine(H, W, 1)
    a = (_t(t[None]), _t(sr[None]), _t(dt[None, None]), _t(ds[None, None]), _t(K[None]), _t(pose[None]))
    for eps in (1e-3, 1e-2, 1e-5):
        L1 = e.linearize(*a, default_opts(w_dc=0.15, irls_eps=eps)); L0 = e.linearize(*a, default_opts(w_dc=0.0, irls_eps=eps))
        O1 = orc.linearize(t, sr, dt, ds, pose, K, oopts(w_dc=0.15, irls_eps=eps)); O0 = orc.linearize(t, sr, dt, ds, pose, K, oopts(w_dc=0.0, irls_eps=eps))
        Hd_e, Hd_o = L1["H"][0] - L0["H"][0], O1["H"] - O0["H"]
        gd_e, gd_o = L1["g"][0] - L0["g"][0], O1["g"] - O0["g"]
        print(f"pair {n} eps {eps}: photo H rel {np.abs(L0['H'][0] - O0['H']).max() / np.abs(O0['H']).max():.2e}  DC H rel {np.abs(Hd_e - Hd_o).max() / np.abs(Hd_o).max():.2e}  "
              f"|H_dc|/|H_photo| {np.abs(Hd_o).max() / np.abs(O0['H']).max():.2e}  DC g rel {np.abs(gd_e - gd_o).max() / np.abs(gd_o).max():.2e}  cost_dc {L1['cost_dc'][0]:.6e} {O1['cost_dc']:.6e}")
        print("   DC H diag rel", (np.diag(Hd_e) - np.diag(Hd_o)) / np.diag(Hd_o))
    ph = orc.photometric(t, sr, dt, ds, pose, K)
    dd = 1 - ph["weight"]
    print("   dd quantiles", np.quantile(dd, [0.01, 0.1, 0.5, 0.9, 0.99]), "frac dd<1e-3", (dd < 1e-3).mean(), "frac dd == 0", (dd == 0).mean())
