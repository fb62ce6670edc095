#!/usr/bin/env python3
"""Secondary timings for DESIGN.md (NOT the bench.py metric): the other BASELINE.json configs on one GPU.
   #2 B=1 pose, #3 8 windows/GPU (64 over 8 GPUs), #4 pose+scale 8 iters, #5 dense 320x240 and 448x256."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import os as _os; _os.environ.setdefault("TCSFM_SET_ENV_DEFAULTS", "1")      # (a measurement script owns its process: HIP_FORCE_DEV_KERNARG / GPU_MAX_HW_QUEUES when absent)
from tightly_coupled_sfm_amd import synth
from tightly_coupled_sfm_amd.engine import Engine, default_opts

def run(name, H, W, npairs, opts, dense=False, steps=300, warm=30):
    b = synth.make_batch(npairs, H, W, seed0=0, both_directions=True)
    d = {k: torch.as_tensor(v).cuda().contiguous() for k, v in b.items()}
    e = Engine(H, W, npairs)
    out = torch.empty_like(d["pose_init"])
    def step():
        if dense:
            e.refine_dense(d["tgt"], d["src"], d["depth_t"], d["depth_s"], d["K"], d["pose_init"], opts)
        else:
            e.refine_into(d["tgt"], d["src"], d["depth_t"], d["depth_s"], d["K"], d["pose_init"], out, opts)
    for _ in range(warm): step()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(steps): step()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / steps
    e.profile_begin()
    for _ in range(50): step()
    pr = e.profile_end()
    lin_us = pr["linearize"][0] / max(pr["linearize"][1], 1) * 1e3
    bpp = 36 + (64 if dense else 0)
    print(json.dumps({"config": name, "HxW": f"{H}x{W}", "directed_pairs": npairs, "iters": opts.n_iters, "us_per_call": round(dt * 1e6, 1),
                      "windows_per_s": round(npairs / 2 / dt, 1), "linearize_us": round(lin_us, 2),
                      "linearize_GBps_algorithmic32": round(32 * H * W * npairs / lin_us / 1e3, 1)}))

run("#2 B=1 pose", 192, 640, 2, default_opts(n_iters=4))
run("#2 B=1 pose + depth-consistency term (reference default l_depth_consist)", 192, 640, 2, default_opts(n_iters=4, w_dc=0.15))
run("#3 8 windows per GPU pose", 192, 640, 16, default_opts(n_iters=4))
run("#3' 64 windows on ONE GPU pose", 192, 640, 128, default_opts(n_iters=4), steps=50, warm=5)
run("#4 pose+scale 8 iters", 192, 640, 2, default_opts(n_iters=8, refine=1))
run("#5 dense 320x240", 240, 320, 2, default_opts(n_iters=4, min_depth=0.03, max_depth=3.0), dense=True)
run("#5 dense 448x256", 256, 448, 2, default_opts(n_iters=4, min_depth=0.03, max_depth=3.0), dense=True)
run("dense 640x192", 192, 640, 2, default_opts(n_iters=4), dense=True)


def run_scale_merged(name, opts, calls=10, streams=2, steps=400, distinct=12):
    """BASELINE config 4 (pose + log depth-scale, 7 x 7) as QUEUED calls merged by the library (tcsfm_refine_window_scale_queued)"""
    H, W = 192, 640
    ws = []
    for i in range(distinct):
        b = synth.make_batch(2, H, W, seed0=31 * i, both_directions=True)
        d = {k: torch.as_tensor(v).cuda().contiguous() for k, v in b.items()}
        ws.append(dict(tgt=d["tgt"][:1].contiguous(), srcs=d["src"][:1][None].contiguous(), dt=d["depth_t"][:1].contiguous(), ds=d["depth_s"][:1][None].contiguous(),
                       pose=d["pose_init"].contiguous(), po=torch.empty(2, 6, device="cuda"), lo=torch.empty(2, device="cuda")))
    K = torch.as_tensor(synth.make_batch(1, H, W)["K"][:1]).cuda().contiguous()
    torch.cuda.synchronize()
    e = Engine(H, W, 2 * calls, lanes=max(2, streams))
    want = [e.refine_window(w["tgt"], w["srcs"], w["dt"], w["ds"], K, w["pose"], opts)[:2] for w in ws]
    want = [(p.clone(), l.clone()) for p, l in want]
    e.set_coalesce(calls); e.set_coalesce_lanes(streams)
    def block(n):
        for k in range(n):
            w = ws[k % distinct]
            e.refine_window_scale_queued(w["tgt"], w["srcs"], w["dt"], w["ds"], K, w["pose"], None, w["po"], w["lo"], opts)
        e.flush(); torch.cuda.synchronize()
    block(60)
    same = all(torch.equal(w["po"], x[0]) and torch.equal(w["lo"], x[1]) for w, x in zip(ws, want))
    t0 = time.perf_counter(); block(steps); dt_ = (time.perf_counter() - t0) / steps
    print(json.dumps({"config": name, "HxW": f"{H}x{W}", "directed_pairs": 2, "iters": opts.n_iters, "calls_per_sequence": calls, "streams": streams,
                      "us_per_window": round(dt_ * 1e6, 1), "windows_per_s": round(1 / dt_, 1), "same_results_as_single_calls": same}))
    e.set_coalesce_lanes(1); e.set_coalesce(0); e.close()


run_scale_merged("#4 pose+scale 8 iters as QUEUED calls merged by the library, 10 per launch sequence on 2 streams", default_opts(n_iters=8, refine=1))


def run_window(name, opts, dense=False, steps=300, warm=30):
    """the reference's KITTI window as ONE call: B=1 target, S=2 sources (4 directed pairs), min over the sources"""
    H, W, B, S = 192, 640, 1, 2
    b = synth.make_batch(2 * S, H, W, seed0=0)
    d = {k: torch.as_tensor(v).cuda().contiguous() for k, v in b.items()}
    tgt, srcs = d["tgt"][:1], d["src"][:S].reshape(S, B, 3, H, W)
    dt, ds = d["depth_t"][:1], d["depth_s"][:S].reshape(S, B, 1, H, W)
    pose = torch.cat([d["pose_init"][:S], -d["pose_init"][:S]]).contiguous()
    e = Engine(H, W, 2 * S * B)
    f = (lambda: e.refine_dense_window(tgt, srcs, dt, ds, d["K"][:1], pose, opts, argmin=True)) if dense else \
        (lambda: e.refine_window(tgt, srcs, dt, ds, d["K"][:1], pose, opts, argmin=True))
    for _ in range(warm): f()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(steps): f()
    torch.cuda.synchronize(); dt_ = (time.perf_counter() - t0) / steps
    print(json.dumps({"config": name, "HxW": f"{H}x{W}", "directed_pairs": 4, "iters": opts.n_iters, "us_per_call": round(dt_ * 1e6, 1), "windows_per_s": round(1 / dt_, 1)}))


def run_window_merged(name, opts, calls=5, streams=2, steps=400, distinct=12):
    """the same KITTI windows as QUEUED calls merged by the library (tcsfm_refine_window_queued, `calls` per launch sequence, the sequences
    alternating over `streams` streams of the handle); distinct windows, per window the bits of the single call (checked)"""
    H, W, B, S = 192, 640, 1, 2
    ws = []
    for i in range(distinct):
        b = synth.make_batch(2 * S, H, W, seed0=40 * i)
        d = {k: torch.as_tensor(v).cuda().contiguous() for k, v in b.items()}
        ws.append(dict(tgt=d["tgt"][:1].contiguous(), srcs=d["src"][:S].reshape(S, B, 3, H, W).contiguous(), dt=d["depth_t"][:1].contiguous(),
                       ds=d["depth_s"][:S].reshape(S, B, 1, H, W).contiguous(), pose=torch.cat([d["pose_init"][:S], -d["pose_init"][:S]]).contiguous(),
                       out=torch.empty(2 * S, 6, device="cuda")))
    K = torch.as_tensor(synth.make_batch(1, H, W, seed0=0)["K"][:1]).cuda().contiguous()
    torch.cuda.synchronize()
    e = Engine(H, W, 2 * S * B * calls, lanes=max(2, streams))
    want = [e.refine_window(w["tgt"], w["srcs"], w["dt"], w["ds"], K, w["pose"], opts, argmin=True)[0].clone() for w in ws]
    e.set_coalesce(calls); e.set_coalesce_lanes(streams)
    def block(n):
        for k in range(n):
            w = ws[k % distinct]
            e.refine_window_queued(w["tgt"], w["srcs"], w["dt"], w["ds"], K, w["pose"], w["out"], opts)
        e.flush(); torch.cuda.synchronize()
    block(60)
    same = all(torch.equal(w["out"], x) for w, x in zip(ws, want))
    t0 = time.perf_counter(); block(steps); dt_ = (time.perf_counter() - t0) / steps
    print(json.dumps({"config": name, "HxW": f"{H}x{W}", "directed_pairs": 4, "iters": opts.n_iters, "calls_per_sequence": calls, "streams": streams,
                      "us_per_window": round(dt_ * 1e6, 1), "windows_per_s": round(1 / dt_, 1), "same_poses_as_single_calls": same}))
    e.set_coalesce_lanes(1); e.set_coalesce(0); e.close()


run_window("KITTI window B=1 S=2, depth consistency, min over sources (window rule PAIR)", default_opts(n_iters=4, w_dc=0.15))
run_window_merged("KITTI windows QUEUED and merged by the library, 5 per launch sequence on 2 streams (window rule PAIR)", default_opts(n_iters=4, w_dc=0.15, argmin=1))
run_window_merged("KITTI windows QUEUED and merged by the library, 8 per launch sequence on 2 streams (window rule PAIR)", default_opts(n_iters=4, w_dc=0.15, argmin=1), calls=8)
run_window("KITTI window B=1 S=2, depth consistency, min over sources (window rule REFERENCE)", default_opts(n_iters=4, w_dc=0.15, window_rule=1))
run_window("KITTI window B=1 S=2 without the depth-consistency term", default_opts(n_iters=4))
run_window("KITTI window dense JOINT (shared depth, 12x12)", default_opts(n_iters=4, dense_joint=1), dense=True)
run_window("KITTI window dense per-pair copies", default_opts(n_iters=4, dense_joint=0), dense=True)


def run_host(steps=200, warm=20):
    """PCIe-inclusive variant of #2: the caller hands over HOST arrays (opts.host_ptrs = 1); never the bench.py metric"""
    import ctypes as C
    H, W, npairs = 192, 640, 2
    b = synth.make_batch(npairs, H, W, seed0=0, both_directions=True)
    host = {k: np.ascontiguousarray(b[k], dtype=np.float32) for k in ("tgt", "src", "depth_t", "depth_s", "K", "pose_init")}
    ptr = lambda a: a.ctypes.data_as(C.c_void_p)
    e = Engine(H, W, npairs)
    o = default_opts(n_iters=4, host_ptrs=1)
    out = np.zeros((npairs, 6), np.float32)
    def step():
        e._call(e.lib.tcsfm_refine(e._h, C.byref(o), npairs, ptr(host["tgt"]), ptr(host["src"]), ptr(host["depth_t"]), ptr(host["depth_s"]),
                                   ptr(host["K"]), ptr(host["pose_init"]), None, ptr(out), None, None))
    for _ in range(warm): step()
    t0 = time.perf_counter()
    for _ in range(steps): step()
    dt = (time.perf_counter() - t0) / steps
    mb = sum(host[k].nbytes for k in ("tgt", "src", "depth_t", "depth_s")) / 1e6
    print(json.dumps({"config": "#2 B=1 pose, HOST pointers (PCIe-inclusive, synchronous, pageable memory)", "us_per_call": round(dt * 1e6, 1),
                      "windows_per_s": round(1 / dt, 1), "MB_in_per_call": round(mb, 2)}))
    # the same with PINNED host arrays (what a caller who cares would hand over)
    pinned = {k: torch.from_numpy(v).pin_memory() for k, v in host.items()}
    host = {k: v.numpy() for k, v in pinned.items()}
    outp = torch.zeros((npairs, 6)).pin_memory(); out = outp.numpy()
    for _ in range(warm): step()
    t0 = time.perf_counter()
    for _ in range(steps): step()
    dt = (time.perf_counter() - t0) / steps
    print(json.dumps({"config": "#2 B=1 pose, HOST pointers, pinned memory", "us_per_call": round(dt * 1e6, 1), "windows_per_s": round(1 / dt, 1)}))

run_host()


def run_window(steps=300, warm=30):
    """the reference's KITTI default as a window: B=1 target, S=2 sources -> 4 directed pairs, min over sources, depth consistency"""
    H, W, B, S = 192, 640, 1, 2
    b = synth.make_batch(2 * S, H, W, seed0=0)
    d = {k: torch.as_tensor(v).cuda().contiguous() for k, v in b.items()}
    tgt, srcs = d["tgt"][:1], d["src"][:S].reshape(S, B, 3, H, W)
    dt, ds = d["depth_t"][:1], d["depth_s"][:S].reshape(S, B, 1, H, W)
    pose = torch.cat([d["pose_init"][:S], -d["pose_init"][:S]]).contiguous()
    e = Engine(H, W, 2 * S * B)
    for argmin in (False, True):
        o = default_opts(n_iters=4, w_dc=0.15)
        for _ in range(warm): e.refine_window(tgt, srcs, dt, ds, d["K"][:1], pose, o, argmin=argmin)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(steps): e.refine_window(tgt, srcs, dt, ds, d["K"][:1], pose, o, argmin=argmin)
        torch.cuda.synchronize(); dt_ = (time.perf_counter() - t0) / steps
        print(json.dumps({"config": f"KITTI-default window B=1 S=2 (4 directed pairs), depth consistency, min over sources={argmin}",
                          "us_per_call": round(dt_ * 1e6, 1), "windows_per_s": round(1 / dt_, 1)}))

run_window()
