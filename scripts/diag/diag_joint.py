"""diagnostic: the pixel(s) where the joint dense mode and its replayed oracle disagree"""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import parity_util as PU
from oracle.oracle import Oracle, default_opts as oopts
import os as _os; _os.environ.setdefault("TCSFM_SET_ENV_DEFAULTS", "1")      # (a measurement script owns its process: HIP_FORCE_DEV_KERNARG / GPU_MAX_HW_QUEUES when absent)
from tightly_coupled_sfm_amd.engine import Engine, default_opts
from test_gpu_joint_dense import _window, _t
orc = Oracle("f64")
B, S, H, W = 1, 2, 192, 640
rule, argmin = 0, True
w = _window(B, S, H, W)
N = 2 * S * B; SB = S * B
for nit in (1, 2, 3, 4):
    e = Engine(H, W, N)
    o = default_opts(n_iters=nit, w_dc=0.0, min_depth=0.06, max_depth=2.67, window_rule=0)
    e.trace_begin(nit, N)
    pose, depth, st = e.refine_dense_window(*(_t(w[k]) for k in ("target", "sources", "depth_t", "depth_s", "K", "first")), o, stats=True, argmin=argmin)
    bits, dec = e.trace_end()
    depth = depth.cpu().numpy()[:, 0]
    pf, df, sf = orc.refine_dense_joint(w["target"], w["sources"], w["depth_t"][:, 0], w["depth_s"][:, :, 0], w["K"], w["first"][:SB], oopts(n_iters=nit),
                                        argmin=argmin, rule=rule, bits=bits[:, :SB], decide=dec[:, :SB], lambda_depth=1.0, w_prior=10.0, min_depth=0.06, max_depth=2.67)
    for b in range(B):
        rel = np.abs(depth[b].astype(np.float64) / df[b] - 1)
        bad = np.argwhere(rel > 2e-5)
        print(f"nit {nit} target {b}: max rel {rel.max():.2e}, {len(bad)} px above 2e-5", [tuple(x) for x in bad[:6]])
        for (v, u) in bad[:3]:
            for s in range(S):
                m = s * B + b
                ix, iy = orc.sample_positions(df[b], pf[m], w["K"][b])
                print(f"    px ({v},{u}) pair {m}: bits per lin", [hex(int(bits[k, m, v, u])) for k in range(nit)], "rel", rel[v, u], "final sample pos", ix[v, u], iy[v, u])

