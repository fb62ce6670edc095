#!/usr/bin/env python3
"""dense window mode on the reference's loss (window_rule REFERENCE) against the library's joint / per-pair dense modes: time per window
(one call in flight, device pointers), per size and source count; round 5: the same windows as QUEUED calls merged through the pointer
table (tcsfm_refine_dense_window_queued, per-call normaliser groups) and the reference's own minibatch of 6 windows per call
(run_sequential_optimization.py:186) -> one JSON line each (profiles/r05_dense_ref_timing.jsonl)"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
import os as _os; _os.environ.setdefault("TCSFM_SET_ENV_DEFAULTS", "1")      # (a measurement script owns its process: HIP_FORCE_DEV_KERNARG / GPU_MAX_HW_QUEUES when absent)
from tightly_coupled_sfm_amd import _lib
from tightly_coupled_sfm_amd.engine import Engine, default_opts
import test_gpu_dense_reference as T

ALG = 36          # algorithmic bytes per pixel, directed pair and iteration of a pose + depth refinement
ONLY = sys.argv[1] if len(sys.argv) > 1 else ""


def modes(mind, maxd):
    kw = dict(n_iters=4, min_depth=mind, max_depth=maxd)
    R = dict(window_rule=_lib.WINDOW_REFERENCE, **kw)
    return (("reference loss (forward + inverse + depth consistency + SSIM prior)", "ref", default_opts(w_dc=0.15, prior_init=0.1, **R)),
            ("reference loss, QUARTER-resolution unknown (the reference's parametrisation, optimizer.py:194-198)", "ref_q", default_opts(w_dc=0.15, prior_init=0.1, depth_param=_lib.DEPTH_QUARTER, **R)),
            ("reference loss, the SOURCE depth maps unknowns too (free_source_depths; every leaf of optimize_depth_pred moves)", "ref_free", default_opts(w_dc=0.15, prior_init=0.1, free_source_depths=1, **R)),
            ("reference loss, the reference's complete leaf set: QUARTER-resolution maps of target AND sources (depth_param QUARTER + free_source_depths)", "ref_q_free",
             default_opts(w_dc=0.15, prior_init=0.1, depth_param=_lib.DEPTH_QUARTER, free_source_depths=1, **R)),
            ("reference loss, forward + inverse only", "ref_fi", default_opts(w_dc=0.0, prior_init=0.0, **R)),
            ("library joint / pair dense mode (own weights, Tikhonov prior)", "lib", default_opts(**kw)))


def emit(**kw):
    print(json.dumps(kw), flush=True)


for H, W, S, mind, maxd in ((240, 320, 1, 0.03, 3.0), (256, 448, 1, 0.03, 3.0), (192, 640, 2, 0.06, 2.67)):
    N1 = 2 * S
    alg_bytes = ALG * H * W * N1 * 4          # per window: 2 S directed pairs x 4 iterations
    # ---- one B = 1 call in flight
    w = T._window(1, S, H, W, seed=31)
    t = {k: T._dev(v) for k, v in w.items()}
    dt4, ds5 = t["depth_t"][:, None].contiguous(), t["depth_s"][:, :, None].contiguous()
    e = Engine(H, W, N1)
    for name, tag, o in modes(mind, maxd):
        if ONLY and ONLY not in tag:
            continue
        step = lambda: e.refine_dense_window(t["tgt"], t["srcs"], dt4, ds5, t["K"], t["pose"], o, argmin=True)
        for _ in range(20): step()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(200): step()
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 200
        e.profile_begin()
        for _ in range(30): step()
        pr = e.profile_end()
        lin = pr["linearize_kernel"][0] / max(pr["linearize_kernel"][1], 1) * 1e3
        emit(HxW=f"{H}x{W}", S=S, windows_per_call=1, launch="one call in flight", mode=name, tag=tag, us_per_window=round(dt * 1e6, 1), windows_per_s=round(1 / dt, 1),
             joint_kernel_us=round(lin, 2), achieved_GBps=round(alg_bytes / dt / 1e9, 1), frac_of_8TBps=round(alg_bytes / dt / 8e12, 4))
    e.close()
    # ---- the same B = 1 windows as QUEUED calls, merged 8 at a time over two streams (bit-identical per window: tests/test_gpu_coalesce.py)
    NW = 16
    ws = [T._window(1, S, H, W, seed=31 + 3 * i) for i in range(NW)]
    ts = [{k: T._dev(v) for k, v in x.items()} for x in ws]
    for x in ts:
        x["dt4"], x["ds5"] = x["depth_t"][:, None].contiguous(), x["depth_s"][:, :, None].contiguous()
    po = [torch.zeros(N1, 6, device="cuda") for _ in range(NW)]
    do = [torch.zeros(N1, 1, H, W, device="cuda") for _ in range(NW)]
    e = Engine(H, W, N1 * 8, lanes=2)
    e.set_coalesce(8); e.set_coalesce_lanes(2)
    for name, tag, o in modes(mind, maxd):
        if tag in ("ref_free", "ref_q_free") or (ONLY and ONLY not in tag):
            continue
        oq = o; oq.argmin = 1
        def rnd():
            for x, p, d in zip(ts, po, do):
                e.refine_dense_window_queued(x["tgt"], x["srcs"], x["dt4"], x["ds5"], x["K"], x["pose"], p, d, oq)
        for _ in range(3): rnd()
        e.synchronize(); t0 = time.perf_counter()
        R = 20
        b0, c0 = e.coalesce_counts()
        for _ in range(R): rnd()
        e.synchronize(); dt = (time.perf_counter() - t0) / (R * NW)
        b1, c1 = e.coalesce_counts()
        per_seq = (c1 - c0) / max(b1 - b0, 1)         # calls per launch sequence actually issued (the library's own dense mode merges only with one source per target)
        launch = (f"queued calls, {per_seq:.0f} per merged sequence, 2 streams, {NW} distinct windows" if b1 > b0 else
                  f"queued calls that do NOT merge in this mode (each runs at once), {NW} distinct windows")
        emit(HxW=f"{H}x{W}", S=S, windows_per_call=1, launch=launch, mode=name, tag=tag, us_per_window=round(dt * 1e6, 1),
             windows_per_s=round(1 / dt, 1), achieved_GBps=round(alg_bytes / dt / 1e9, 1), frac_of_8TBps=round(alg_bytes / dt / 8e12, 4))
    e.set_coalesce_lanes(1); e.set_coalesce(0)
    e.close()
    # ---- the reference's minibatch: 6 windows per call (its batch normalisers couple them)
    B = 6
    w = T._window(B, S, H, W, seed=31)
    t = {k: T._dev(v) for k, v in w.items()}
    dt4, ds5 = t["depth_t"][:, None].contiguous(), t["depth_s"][:, :, None].contiguous()
    e = Engine(H, W, N1 * B)
    for name, tag, o in modes(mind, maxd):
        if ONLY and ONLY not in tag:
            continue
        step = lambda: e.refine_dense_window(t["tgt"], t["srcs"], dt4, ds5, t["K"], t["pose"], o, argmin=True)
        for _ in range(10): step()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(60): step()
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 60
        emit(HxW=f"{H}x{W}", S=S, windows_per_call=B, launch="one minibatch-6 call in flight (run_sequential_optimization.py:186)", mode=name, tag=tag,
             us_per_call=round(dt * 1e6, 1), us_per_window=round(dt * 1e6 / B, 1), windows_per_s=round(B / dt, 1),
             achieved_GBps=round(B * alg_bytes / dt / 1e9, 1), frac_of_8TBps=round(B * alg_bytes / dt / 8e12, 4))
    e.close()
