#!/usr/bin/env python3
"""Which scalar goes down faster: the reference's compute_optimization_loss (optimizer.py:47-86) evaluated at the poses
that k Gauss-Newton iterations reach under the PAIR rule and under the REFERENCE rule (float64 oracle, CPU).
    python scripts/experiments/window_rule_study.py  ->  profiles/r03_window_rule_study.json"""
import json, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import standins
from oracle.oracle import Oracle, default_opts as oopts
orc = Oracle("f64")
out = []
for (B, S, H, W, seed) in ((2, 2, 96, 320, 90), (1, 2, 96, 320, 7), (2, 3, 48, 160, 11)):
    w = standins.make_window(B, S, H, W, seed0=seed)
    w["depth_t"] = orc.disp_to_depth(w["disp_t"], 0.06, 2.67)[1].astype(np.float32)
    w["depth_s"] = orc.disp_to_depth(w["disp_s"], 0.06, 2.67)[1].astype(np.float32)
    a = (w["target"], w["sources"], w["depth_t"][:, 0], w["depth_s"][:, :, 0], w["K"])
    ref_loss = lambda p: float(orc.linearize_window(*a, p, oopts(n_iters=1, w_dc=0.15), argmin=True, rule=1)["cost"].sum())
    row = {"window": f"B={B} S={S} {W}x{H}", "reference_loss_at_start": ref_loss(w["first"]), "by_rule": {}}
    for rule in (0, 1):
        for solver in (0, 1):
            tr = []
            for k in (1, 2, 4, 8, 16):
                p, _, _ = orc.refine_window(*a, w["first"], oopts(n_iters=k, w_dc=0.15, solver=solver, lambda0=1e-4 if solver == 0 else 1e-3), argmin=True, rule=rule)
                tr.append(round(ref_loss(p), 6))
            row["by_rule"][f"rule{rule}_{'lm' if solver else 'gn'}"] = tr
    out.append(row)
    print(json.dumps(row))
json.dump({"what": "reference loss (compute_optimization_loss, default options) after k = 1, 2, 4, 8, 16 iterations under each window rule", "rows": out},
          open(os.path.join(ROOT, "profiles", "r03_window_rule_study.json"), "w"), indent=1)
