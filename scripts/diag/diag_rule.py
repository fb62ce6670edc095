"""diagnostic: per-pair pose / cost errors of the window refinement vs the replayed oracle under both window rules"""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import parity_util as PU
from oracle.oracle import Oracle, default_opts as oopts
import os as _os; _os.environ.setdefault("TCSFM_SET_ENV_DEFAULTS", "1")      # (a measurement script owns its process: HIP_FORCE_DEV_KERNARG / GPU_MAX_HW_QUEUES when absent)
from tightly_coupled_sfm_amd.engine import Engine, default_opts
from test_gpu_window_rule import _window, _t
orc = Oracle("f64")
B, S, H, W = 2, 2, 96, 320
w = _window(B, S, H, W)
N = 2 * S * B
for rule in (0, 1):
    for w_dc in (0.0, 0.15, 0.6):
        for nit in (1, 2, 4):
            e = Engine(H, W, N)
            o = default_opts(window_rule=rule, w_dc=w_dc, n_iters=nit)
            args = tuple(_t(w[k]) for k in ("target", "sources", "depth_t", "depth_s", "K", "first"))
            e.trace_begin(nit, N)
            pose, _, st = e.refine_window(*args, o, stats=True, argmin=True)
            bits, dec = e.trace_end()
            pose = pose.cpu().numpy().astype(np.float64); st = st.cpu().numpy()
            rp, _, rst = orc.refine_window(w["target"], w["sources"], w["depth_t"][:, 0], w["depth_s"][:, :, 0], w["K"], w["first"],
                                           oopts(w_dc=w_dc, n_iters=nit), argmin=True, bits=bits, decide=dec, rule=rule)
            errs = [PU.pose_err(pose[n], rp[n]) for n in range(N)]
            ce = np.abs(st[:, :nit, 0] - rst[:, :nit, 0]) / rst[:, :nit, 0]
            print(f"rule {rule} w_dc {w_dc} nit {nit}: max t-err {max(a for a, _ in errs):.2e} r-err {max(b for _, b in errs):.2e} (pair {int(np.argmax([b for _, b in errs]))})  cost err {ce.max():.2e}", flush=True)
# one linearisation: H and g of pair 7 vs the oracle under rule 1
e = Engine(H, W, N)
o = default_opts(window_rule=1, w_dc=0.15, n_iters=1)
L = e.linearize_window(*(_t(w[k]) for k in ("target", "sources", "depth_t", "depth_s", "K", "first")), o, argmin=True)
Lo = orc.linearize_window(w["target"], w["sources"], w["depth_t"][:, 0], w["depth_s"][:, :, 0], w["K"], w["first"], oopts(w_dc=0.15, n_iters=1), argmin=True, rule=1)
for n in range(N):
    print(n, "g rel", np.abs(L["g"][n] - Lo["g"][n]).max() / np.abs(Lo["g"][n]).max(), "H rel", np.abs(L["H"][n] - Lo["H"][n]).max() / np.abs(Lo["H"][n]).max(),
          "cost", L["cost"][n], Lo["cost"][n], "nmask", L["n_mask"][n], Lo["n_mask"][n])
