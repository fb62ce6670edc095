#!/usr/bin/env python3
"""A/B helper (scripts/experiments/ab_modes.sh): BASELINE config 4 (pose + depth scale, k_linearize<7>) with whatever library is in place --
one call at a time, a 64-pair call, queued calls merged ten at a time -- plus a hash of the refined poses / scales of fixed inputs in the
6-DoF, 7-DoF and depth-consistency modes, so that builds that must be bit-identical can be told apart from builds that are not."""
import hashlib, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import os as _os; _os.environ.setdefault("TCSFM_SET_ENV_DEFAULTS", "1")      # (a measurement script owns its process: HIP_FORCE_DEV_KERNARG / GPU_MAX_HW_QUEUES when absent)
from tightly_coupled_sfm_amd import synth
from tightly_coupled_sfm_amd.engine import Engine, default_opts

H, W = 192, 640
def batch(n, seed0=0):
    b = synth.make_batch(n, H, W, seed0=seed0, both_directions=True)
    return {k: torch.as_tensor(v).cuda().contiguous() for k, v in b.items()}

def timed(fn, steps, warm):
    for _ in range(warm): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(steps): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / steps

out = {}
h = hashlib.sha256()
for name, kw in (("pose6", dict(n_iters=4)), ("pose6_dc", dict(n_iters=4, w_dc=0.15)), ("scale7", dict(n_iters=8, refine=1)), ("scale7_dc", dict(n_iters=8, refine=1, w_dc=0.15)),
                 ("pose6_lm", dict(n_iters=6, solver=1))):
    d = batch(4, 7)
    e = Engine(H, W, 4)
    r = e.refine(d["tgt"], d["src"], d["depth_t"], d["depth_s"], d["K"], d["pose_init"], default_opts(**kw))
    torch.cuda.synchronize()
    for t in r:
        if torch.is_tensor(t): h.update(t.detach().cpu().numpy().tobytes())
    e.close()
out["results_sha"] = h.hexdigest()[:16]

opts = default_opts(n_iters=8, refine=1)
for npairs, steps, warm in ((2, 300, 30), (128, 30, 5)):
    d = batch(npairs)
    e = Engine(H, W, npairs)
    o = torch.empty_like(d["pose_init"])
    fn = lambda: e.refine_into(d["tgt"], d["src"], d["depth_t"], d["depth_s"], d["K"], d["pose_init"], o, opts)
    dt = timed(fn, steps, warm)
    e.profile_begin()
    for _ in range(20): fn()
    pr = e.profile_end()
    out[f"scale7_pairs{npairs}"] = {"us_per_call": round(dt * 1e6, 1), "linearize_us": round(pr["linearize"][0] / max(pr["linearize"][1], 1) * 1e3, 2)}
    e.close()
# the reference's KITTI window (B targets, S = 2 sources, min over the sources, depth consistency): k_linearize<SEL>
def window(B, steps, warm, rule=0):
    S = 2
    b = synth.make_batch(2 * S * B, H, W, seed0=3)
    d = {k: torch.as_tensor(v).cuda().contiguous() for k, v in b.items()}
    tgt, srcs = d["tgt"][:B].contiguous(), d["src"][:S * B].reshape(S, B, 3, H, W).contiguous()
    dt, ds = d["depth_t"][:B].contiguous(), d["depth_s"][:S * B].reshape(S, B, 1, H, W).contiguous()
    pose = torch.cat([d["pose_init"][:S * B], -d["pose_init"][:S * B]]).contiguous()
    e = Engine(H, W, 2 * S * B)
    o = default_opts(n_iters=4, w_dc=0.15, window_rule=rule)
    r = e.refine_window(tgt, srcs, dt, ds, d["K"][:B].contiguous(), pose, o, argmin=True)
    torch.cuda.synchronize()
    hh = hashlib.sha256()
    for t in r:
        if torch.is_tensor(t): hh.update(t.detach().cpu().numpy().tobytes())
    dt_ = timed(lambda: e.refine_window(tgt, srcs, dt, ds, d["K"][:B].contiguous(), pose, o, argmin=True), steps, warm)
    e.close()
    return {"us_per_call": round(dt_ * 1e6, 1), "sha": hh.hexdigest()[:12]}
out["kitti_window_B1"] = window(1, 300, 30)
out["kitti_window_B8"] = window(8, 60, 6)
out["kitti_window_B1_reference_rule"] = window(1, 200, 20, rule=1)
print(json.dumps(out))
