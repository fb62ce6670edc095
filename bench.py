#!/usr/bin/env python3
"""bench.py -- optimised frame-pairs/sec at 640x192, 4 GN iterations (BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--windows-per-gpu B]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

`python bench.py --gpus N` with N > 1 and no WORLD_SIZE in the environment starts itself: the parent process, which never touches
the GPU, launches torch.distributed.run with N ranks of this file (one per GPU, rendezvous on 127.0.0.1) and exits with its code.

Workload.  N = 1 (default): BASELINE.json configs[1] / SURVEY 8d config #2 -- one window of B=1 target frame with S=1 source
frame -> the reference's fwd + inv directed pairs (2 S B = 2, train_mono.py:54-62), 640x192, 4 Gauss-Newton iterations of the
6-DoF pose of each directed pair.  N > 1: configs[2] / config #3 -- 64 windows over 8 GPUs = 8 windows (16 directed pairs) per
rank and step (--windows-per-gpu overrides either default).  One *step* = one ``tcsfm_refine_window`` call over the rank's batch; one
frame-pair is counted per window (not per directed pair).  Inputs are synthetic (tightly_coupled_sfm_amd.synth), resident in HBM
before the timed region.  The steps ROTATE over a ring of distinct windows whose inputs exceed the 256 MiB Infinity Cache (--ring-mb,
default 320 MB: 84 windows at B=1), so the images a step reads were last touched ~80 steps earlier and come from HBM, as in a real
sequence; `hot_cache` repeats the blocks on ONE window (the round-3 protocol: everything cache resident) and says how far `value` moves.

Calls in flight.  Steps are independent windows (as the windows of a sequence are), so the handle keeps --lanes of them (default
4) in flight on its lanes (include/tcsfm.h: own HIP stream and scratch per lane): the kernels of one call fill the gaps between
the short kernels of the other.  Every step is still one B-window ``tcsfm_refine_window`` call; `single_stream` reports the same
blocks with one call in flight (the round-1 protocol).

Merged sequences.  The same steps are also run as QUEUED calls (tcsfm_refine_window_queued): the library merges every --coalesce of them
(default 10) into ONE launch sequence over all their directed pairs; consecutive sequences alternate over --coalesce-lanes streams of the handle
(default 2: the 20-workgroup solve kernels of one sequence overlap the chip-filling launches of the other), and what is still waiting at the
end of a block is launched by the block's closing flush.  Per window the poses are the same bits.  The faster of the two ways is the headline; `launch_mode.merged` / `launch_mode.lanes` report both.

Timing.  W warm-up steps, then blocks of EXACTLY K steps, each bracketed by barrier + torch.cuda.synchronize() on both sides and
reduced with MAX over the ranks.  One block is the contract's measurement; because the driver's K=20 block lasts ~1.5 ms, the
block is repeated (R blocks, >= 50 ms in total, R <= 64) and the MEDIAN block is reported (`timed_blocks`, `ms_per_step_blocks`
carry R and the spread; the first block is in there too).

Multi-GPU: windows are independent least-squares problems -> every rank refines its own windows (weak scaling, no collective on
the data path); one RCCL all_gather of the refined poses AFTER the timed region is the "final gather" of the north star, timed
separately (`final_gather_us`).

The JSON line also carries
  launch_mode   the lanes' calls of the timed region were enqueued either as plain launches (nine per call) or as one captured HIP graph
                per call (tcsfm_set_graph_replay; same kernels, same bits): untimed probe blocks of both modes decide (--graph-replay
                auto), `other_mode` repeats the blocks in the mode that lost, `host_enqueue_us_per_step` is the host's share of a step.
  roofline      dominant kernel (k_linearize): algorithmic bytes (32 B/pixel/pair/iteration) per launch divided by the
                launch's duration MEASURED IN THIS RUN with one call in flight: `frac` / `achieved` / `avg_launch_us` come from the GPU's
                own bracket of every launch (every workgroup stamps s_memrealtime at its start and end, duration = latest end - earliest
                start, tcsfm_profile_kernel_time); `avg_launch_us_hip_events` is the HIP event pair around the same launches on the
                launch stream (2-4 us high on a ~10 us kernel).  `rocprof_committed` is rocprofv3's AverageNs of the kernel in the
                committed --kernel-trace --stats run of this command (profiles/<tag>_lanes1_kernel_stats.csv; it includes ~1.2 us of
                dispatch and completion the in-kernel bracket does not see); `frac_consistent` is false when the two differ by more
                than 15 % after that offset -- a stale profile or a regressed kernel -- and `profiles_match_source` says whether the
                profiles were collected on the kernel sources of this tree (hash in profiles/<tag>_meta.json).  `traffic` = HBM bytes
                per launch of THIS workload from the committed PMC passes (profiles/<tag>_pmc_traffic.json).  `bound_actual` = "valu":
                the bound the kernel really runs into is VALU issue (`valu_bound`: its instruction stream priced with measured
                per-class issue costs, profiles/<tag>_valu_census.json); `whole_call` relates the traffic of all nine launches of a
                call to its algorithmic bytes.
  cpu_baseline  the float64 CPU oracle (a scalar C port of the same algorithm, oracle/tcsfm_oracle.c) timed on this box's host
                cores on a bounded sample of the same workload: all cores of the box's share (`value`) and one core (`one_thread`),
                plus the reference's own style of step (PyTorch autograd + Adam, oracle/torch_twin.py).
"""
import argparse
import glob
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

H, W, ITERS = 192, 640, 4
SOURCES = 1                   # S
HBM_PEAK_GBPS = 8000.0        # MI355X_MICROARCH.md: HBM3E 8 TB/s spec


def _oracle_rate(seconds, cores):
    """windows/s of the CPU oracle with `cores` threads busy (one window = fwd + inv pair, 4 GN iterations, at a time per thread;
    the ctypes call into the C oracle releases the GIL)"""
    import threading
    from oracle.oracle import Oracle, default_opts
    from tightly_coupled_sfm_amd import synth
    orc = Oracle("f64")
    opts = default_opts(n_iters=ITERS)
    batches = [synth.make_batch(2, H, W, seed0=1000 + i, both_directions=True) for i in range(2)]
    counts = [0] * cores
    t0 = time.perf_counter()
    deadline = t0 + seconds

    def worker(k):
        i = k
        while time.perf_counter() < deadline:
            b = batches[i % len(batches)]
            for n in range(2):
                orc.refine(b["tgt"][n], b["src"][n], b["depth_t"][n, 0], b["depth_s"][n, 0], b["pose_init"][n], b["K"][n], opts)
            counts[k] += 1
            i += 1

    threads = [threading.Thread(target=worker, args=(k,)) for k in range(cores)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    dt = time.perf_counter() - t0
    return sum(counts) / dt, sum(counts), dt, batches


def cpu_baseline(seconds: float):
    """Bounded sample (~`seconds` of wall time in total) of the bench workload on the host cores: the oracle on all cores of this
    box's share and on one core, then the reference-style Adam step."""
    cores = max(1, min(16, len(os.sched_getaffinity(0))))
    rate, done, dt, batches = _oracle_rate(0.55 * seconds, cores)
    rate1, done1, dt1, _ = _oracle_rate(0.25 * seconds, 1)
    out = {"value": rate, "unit": "frame-pairs/s", "cores": cores, "kind": "port",
           "sample": f"{done} windows x (fwd+inv pair) x {ITERS} GN iterations at {W}x{H}, float64 scalar C oracle, "
                     f"{cores} threads (one window each at a time), {dt:.1f} s",
           "one_thread": {"value": rate1, "unit": "frame-pairs/s", "cores": 1, "sample": f"{done1} windows, {dt1:.1f} s"}}
    # the reference's OWN style of optimisation step -- autograd through the PyTorch residual + Adam
    # (oracle/torch_twin.py, a restatement pinned on the reference's golden vectors) -- on the same window, all cores and one
    try:
        import torch
        from oracle import torch_twin
        from tightly_coupled_sfm_amd import synth
        b = batches[0]
        T = lambda a: torch.tensor(a, dtype=torch.float32)
        sig = lambda d: T(synth.depth_to_sigmoid_disp(d.astype("float64")).astype("float32"))
        args = (T(b["tgt"][:1]), T(b["src"][:1]), sig(b["depth_t"][:1]), sig(b["depth_s"][:1]), T(b["K"][:1]), T(b["pose_init"][:1]))
        rate, steps = torch_twin.time_adam_steps(*args, seconds=0.12 * seconds, threads=cores)
        rate1, steps1 = torch_twin.time_adam_steps(*args, seconds=0.08 * seconds, threads=1)
        out["reference_style"] = {"adam_steps_per_s": round(rate, 2), "pair_iters_per_s": round(2 * rate, 2),
                                  "frame_pairs_per_s_at_20_epochs": round(rate / 20, 3), "threads": cores, "steps": steps,
                                  "one_thread": {"adam_steps_per_s": round(rate1, 2), "frame_pairs_per_s_at_20_epochs": round(rate1 / 20, 4), "steps": steps1},
                                  "what": "PyTorch-CPU autograd + Adam step on pose and quarter-resolution disparity of one "
                                          "window (fwd+inv pair), the reference's way of optimising (20 epochs per window)"}
    except Exception as ex:      # the headline baseline above does not depend on this leg
        out["reference_style"] = {"error": repr(ex)}
    return out


# The committed profiles this bench line cites (scripts/collect_profiles.sh <tag> on the GPU box, scripts/summarise_profiles.py <tag>):
# exact file names of the NEWEST tag that has the file -- no globbing (r02's line picked up another mode's PMC file through sorted(glob)[-1]).
PROFILE_TAGS = ("r05", "r04", "r03")
KERNEL = "k_linearize<6, false, 1"       # the S = 1, no-depth-consistency, MODE_LIN instantiation the bench workload runs
DISPATCH_OFFSET_US = 1.2                  # rocprofv3's kernel duration minus the in-kernel bracket (dispatch + completion), measured r02 / r03
SOURCE_FILES = ("tightly_coupled_sfm_amd/csrc", "include/tcsfm.h")


def source_hash():
    """sha256 over the kernel sources: profiles collected on another tree say so (profiles/<tag>_meta.json, scripts/summarise_profiles.py)"""
    import hashlib
    h = hashlib.sha256()
    files = []
    for rel in SOURCE_FILES:
        p = os.path.join(ROOT, rel)
        files += sorted(os.path.join(p, f) for f in os.listdir(p)) if os.path.isdir(p) else [p]
    for f in files:
        if f.endswith((".h", ".hip")):
            h.update(os.path.basename(f).encode()); h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


def _profile(name):
    for tag in PROFILE_TAGS:
        f = os.path.join(ROOT, "profiles", f"{tag}_{name}")
        if os.path.exists(f):
            return f
    return None


def profiles_meta():
    """{tag, source_hash, matches} of the newest committed profile set, or None"""
    f = _profile("meta.json")
    if not f:
        return None
    try:
        m = json.load(open(f))
        return {"file": os.path.relpath(f, ROOT), "source_hash": m.get("source_hash"), "matches": m.get("source_hash") == source_hash()}
    except Exception:
        return None


def load_pmc(key="hbm_bytes_per_linearize_launch"):
    """HBM bytes per k_linearize launch of THIS workload from the separate rocprofv3 --pmc passes of `python bench.py --lanes 1`
    (profiles/<tag>_pmc_traffic.json: FETCH_SIZE x 2 + WRITE_SIZE as MI355X_MICROARCH.md prescribes), or None."""
    f = _profile("pmc_traffic.json")
    if not f:
        return None
    try:
        return json.load(open(f)).get(key)
    except Exception:
        return None


def rocprof_avg_us(kernel_substr=KERNEL):
    """AverageNs of k_linearize in the committed rocprofv3 --kernel-trace --stats runs: of `python bench.py` (default command), of
    `python bench.py --lanes 1` (every launch has the chip: what the roofline compares against) and of the chip-filling workload
    (scripts/sat_workload.py, 32 windows per call)"""
    import csv
    out = {}
    for key, name in (("default_command", "kernel_stats.csv"), ("lanes_1", "lanes1_kernel_stats.csv"), ("saturated", "sat_kernel_stats.csv")):
        f = _profile(name)
        if not f:
            continue
        try:
            for row in csv.DictReader(open(f)):       # (several instantiations may match -- the shared-pack and the two-pack form: the one the run used most)
                if kernel_substr in row["Name"] and int(row["Calls"]) > out.get(key, {"calls": 0})["calls"]:
                    out[key] = {"us": round(float(row["AverageNs"]) * 1e-3, 3), "calls": int(row["Calls"]), "file": os.path.relpath(f, ROOT)}
        except Exception:
            pass
    return out or None


def consistent(live_us, rocprof_us):
    """the live in-kernel bracket against the committed rocprof average of the same kernel: equal within 15 % after the dispatch offset"""
    return abs(live_us + DISPATCH_OFFSET_US - rocprof_us) <= 0.15 * rocprof_us


def valu_bound(avg_s, pairs_per_launch):
    """VALU issue time of one launch: waves per SIMD x (instructions of one wave priced with the measured per-class issue costs,
    profiles/<tag>_valu_census.json) / shader clock.  The kernel cannot run faster than this whatever the memory system does."""
    f = _profile("valu_census.json")
    if not f:
        return None
    c = json.load(open(f))
    waves = (H * W // 64) * pairs_per_launch                     # one wave per 64 target pixels
    clk = c["predicted_valu_busy_clk_per_wave"] * waves / 1024.0  # per SIMD (256 CUs x 4), perfectly balanced
    ghz = c.get("shader_clock_GHz", 2.4)
    t = clk / (ghz * 1e9)
    return {"valu_insts_per_wave": c["valu_insts_per_wave_census"], "busy_clk_per_wave": c["predicted_valu_busy_clk_per_wave"],
            "bound_us": round(t * 1e6, 3), "frac_of_bound": round(t / avg_s, 4), "clock_GHz": ghz, "census": os.path.relpath(f, ROOT)}


def modes_block(budget_s=12.0):
    """The other BASELINE.json configs and the reference-loss dense mode on ONE GPU, bounded (~`budget_s` seconds in all): per mode
    windows/s, us per call and the roofline fraction of its dominant kernel from THIS run's in-kernel bracket (every workgroup stamps
    s_memrealtime; tcsfm_profile_kernel_time) -- so that the driver's own bench line, not only builder-run scripts, carries a number for
    every config.  Algorithmic bytes per pixel, directed pair and iteration: 32 (pose, pose + scale), 36 (pose + depth)."""
    import numpy as np
    import torch
    from tightly_coupled_sfm_amd import synth, _lib
    from tightly_coupled_sfm_amd.engine import Engine, default_opts
    t_all = time.perf_counter()
    out = {}

    def measure(e, step, pairs, iters, px, bpp, kernel, kernel_pairs, slice_s):
        for _ in range(8):
            step()
        torch.cuda.synchronize()
        n, t0 = 0, time.perf_counter()
        while True:
            for _ in range(10):
                step()
            n += 10
            torch.cuda.synchronize()
            if time.perf_counter() - t0 > slice_s or n >= 4000:
                break
        dt = (time.perf_counter() - t0) / n
        e.profile_begin()
        for _ in range(20):
            step()
        pr = e.profile_end()
        k_us = pr["linearize_kernel"][0] / max(pr["linearize_kernel"][1], 1) * 1e3
        alg_k = bpp * px * kernel_pairs
        r = {"us_per_call": round(dt * 1e6, 1), "calls_per_s": round(1 / dt, 1), "frame_pairs_per_s": round(pairs / 2 / dt, 1), "calls_timed": n,
             "directed_pairs_per_call": pairs, "iters": iters,
             "whole_call": {"algorithmic_bytes": bpp * px * pairs * iters, "achieved_GBps": round(bpp * px * pairs * iters / dt / 1e9, 1),
                            "frac": round(bpp * px * pairs * iters / dt / 1e9 / HBM_PEAK_GBPS, 5)},
             "dominant_kernel": {"kernel": kernel, "avg_launch_us": round(k_us, 2), "algorithmic_bytes_per_launch": alg_k,
                                 "achieved_GBps": round(alg_k / max(k_us, 1e-9) / 1e3, 1), "frac": round(alg_k / max(k_us, 1e-9) / 1e3 / HBM_PEAK_GBPS, 5),
                                 "frac_source": "this run: in-kernel s_memrealtime bracket, one call in flight"}}
        return r

    per = budget_s / 10.0
    # ---- BASELINE config 4: depth scale + 6-DoF pose, 8 iterations, 640x192, one window per call
    b = synth.make_batch(2, H, W, seed0=0, both_directions=True)
    d = {k: torch.as_tensor(v).cuda().contiguous() for k, v in b.items()}
    e = Engine(H, W, 2)
    o = default_opts(n_iters=8, refine=_lib.REFINE_POSE_SCALE)
    outp = torch.empty_like(d["pose_init"])
    out["config4_pose_scale_8it_640x192"] = measure(e, lambda: e.refine_into(d["tgt"], d["src"], d["depth_t"], d["depth_s"], d["K"], d["pose_init"], outp, o),
                                                     2, 8, H * W, 32, "k_linearize<7>", 2, per)
    # ---- the reference's KITTI window as ONE call: B = 1 target, S = 2 sources, min over the sources, depth consistency, rule REFERENCE (pose mode)
    S2 = 2
    bw = synth.make_batch(2 * S2, H, W, seed0=0)
    dw = {k: torch.as_tensor(v).cuda().contiguous() for k, v in bw.items()}
    tgt, srcs = dw["tgt"][:1].contiguous(), dw["src"][:S2].reshape(S2, 1, 3, H, W).contiguous()
    dt_, ds_ = dw["depth_t"][:1].contiguous(), dw["depth_s"][:S2].reshape(S2, 1, 1, H, W).contiguous()
    pose_w = torch.cat([dw["pose_init"][:S2], -dw["pose_init"][:S2]]).contiguous()
    Kw = dw["K"][:1].contiguous()
    e2 = Engine(H, W, 2 * S2)
    ow = default_opts(n_iters=4, w_dc=0.15, window_rule=_lib.WINDOW_REFERENCE)
    out["kitti_window_S2_pose_reference_rule_640x192"] = measure(e2, lambda: e2.refine_window(tgt, srcs, dt_, ds_, Kw, pose_w, ow, argmin=True),
                                                                   2 * S2, 4, H * W, 32, "k_linearize<6, DC, SEL>", 2 * S2, per)
    # ---- the same window, pose + depth on the reference's own loss (optimize_depth_pred: optimizer.py:47-90), full-resolution unknown
    od = default_opts(n_iters=4, w_dc=0.15, prior_init=0.1, window_rule=_lib.WINDOW_REFERENCE)
    dt4, ds5 = dt_[:, None].contiguous() if dt_.dim() == 3 else dt_, ds_
    out["kitti_window_S2_reference_loss_dense_640x192"] = measure(e2, lambda: e2.refine_dense_window(tgt, srcs, dt4, ds5, Kw, pose_w, od, argmin=True),
                                                                    2 * S2, 4, H * W, 36, "k_dense_joint<2, REF> (forward pairs)", S2, per)
    e2.close(); e.close()
    # ---- ... and the reference driver's own shape: a MINIBATCH of 6 such windows in one call (run_sequential_optimization.py:186; the loss couples them
    # through its batch normalisers) -- the chip-filling figure of the mirror's default mode; the joint kernel's bracket is then a chip-full launch
    MB = 6
    bm = synth.make_batch(2 * S2 * MB, H, W, seed0=0)
    dm = {k: torch.as_tensor(v).cuda().contiguous() for k, v in bm.items()}
    tgt6 = dm["tgt"][:MB].contiguous()
    srcs6 = dm["src"][:S2 * MB].reshape(S2, MB, 3, H, W).contiguous()
    dt6 = dm["depth_t"][:MB].contiguous()
    dt6 = dt6[:, None].contiguous() if dt6.dim() == 3 else dt6
    ds6 = dm["depth_s"][:S2 * MB].reshape(S2, MB, 1, H, W).contiguous()
    pose6 = torch.cat([dm["pose_init"][:S2 * MB], -dm["pose_init"][:S2 * MB]]).contiguous()
    K6 = dm["K"][:MB].contiguous()
    e6 = Engine(H, W, 2 * S2 * MB)
    r6 = measure(e6, lambda: e6.refine_dense_window(tgt6, srcs6, dt6, ds6, K6, pose6, od, argmin=True),
                 2 * S2 * MB, 4, H * W, 36, "k_dense_joint<2, REF> (forward pairs of 6 targets: a chip-filling launch)", S2 * MB, per)
    r6["windows_per_call"] = MB
    r6["windows_per_s"] = round(MB * r6["calls_per_s"], 1)
    r6["dominant_kernel"]["frac_source"] = "this run: in-kernel s_memrealtime bracket of the 6-target launches"
    out["kitti_window_S2_reference_loss_dense_minibatch6_640x192"] = r6
    e6.close()
    # ---- BASELINE config 5: per-pixel inverse depth + pose, Schur complement: 320x240 and the reference's own ScanNet size 448x256
    for (hh, ww) in ((240, 320), (256, 448)):
        bb = synth.make_batch(2, hh, ww, seed0=0, both_directions=True)
        dd = {k: torch.as_tensor(v).cuda().contiguous() for k, v in bb.items()}
        ed = Engine(hh, ww, 2)
        ol = default_opts(n_iters=4, min_depth=0.03, max_depth=3.0)
        out[f"config5_dense_schur_{ww}x{hh}"] = measure(ed, lambda: ed.refine_dense(dd["tgt"], dd["src"], dd["depth_t"], dd["depth_s"], dd["K"], dd["pose_init"], ol),
                                                         2, 4, hh * ww, 36, "k_dense_linearize", 2, per)
        # ... and on the reference's loss (window form, S = 1): full-resolution and the reference's quarter-resolution parametrisation
        t1, s1 = dd["tgt"][:1].contiguous(), dd["src"][:1][None].contiguous()
        a1, b1 = dd["depth_t"][:1].contiguous(), dd["depth_s"][:1][None].contiguous()
        a1 = a1[:, None].contiguous() if a1.dim() == 3 else a1
        b1 = b1[:, :, None].contiguous() if b1.dim() == 4 else b1
        p1 = torch.cat([dd["pose_init"][:1], dd["pose_init"][1:2]]).contiguous()
        K1 = dd["K"][:1].contiguous()
        for tag, dp in (("full", _lib.DEPTH_FULL), ("quarter", _lib.DEPTH_QUARTER)):
            orf = default_opts(n_iters=4, w_dc=0.15, prior_init=0.1, min_depth=0.03, max_depth=3.0, window_rule=_lib.WINDOW_REFERENCE, depth_param=dp)
            out[f"config5_reference_loss_{tag}_{ww}x{hh}"] = measure(ed, lambda: ed.refine_dense_window(t1, s1, a1, b1, K1, p1, orf, argmin=True),
                                                                      2, 4, hh * ww, 36, "k_dense_joint<1, REF> (forward pair)", 1, per)
        ed.close()
    out["seconds"] = round(time.perf_counter() - t_all, 1)
    out["what"] = ("one call in flight, device-resident inputs, untimed for the headline; `whole_call` = algorithmic bytes of all directed pairs and "
                   "iterations of a call over its wall time, `dominant_kernel` = this run's in-kernel bracket of the mode's linearisation kernel")
    return out


def shim_block(seconds=4.0):
    """DepthOptimizer.optimize_window (the drop-in for the reference's optimiser, optimizer.py:136-297) end to end with the stand-in
    networks of tests/standins.py, beside the engine call inside it: what a caller of the shim sees per window."""
    import torch
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import standins
    from tightly_coupled_sfm_amd.optimizer import DepthOptimizer
    res = {}
    B, S, ITER = 1, 2, 3
    w = standins.make_window(B, S, H, W)
    cfg = {"minibatch": B, "device": "cuda", "min_depth": 0.06, "max_depth": 2.67, "iterations": ITER, "camera_height": 1.65, "flow_type": "none"}
    for tag, extra in (("pose", {"optimize_depth_pred": False}), ("pose_depth_reference_loss", {"optimize_depth_pred": True})):
        opts = {"epochs": 5, "diff_img_argmin": True, "automasking": True, "mode": "scaled", "l_depth_consist": True, "l_depth_consist_weight": 0.15,
                "l_depth_init": True, "l_depth_init_weight": 0.1, "num_source_imgs": S, "avg_final_epochs": 5}
        opts.update(extra)
        pm, dm = standins.window_models(w, ITER, device="cuda")
        opt = DepthOptimizer(opts, cfg, pm, dm, "09_02")
        data = standins.loader_batch(w, device="cuda")
        for _ in range(4):
            opt.optimize_window(0, data)
        torch.cuda.synchronize()
        n, t0 = 0, time.perf_counter()
        while time.perf_counter() - t0 < seconds / 2 and n < 200:
            opt.optimize_window(0, data)
            n += 1
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / n
        opt.time_engine = True          # the engine call inside it, on its own (device-synchronised around the call: a few extra runs, untimed above)
        calls = []
        for _ in range(8):
            opt.optimize_window(0, data)
            calls.append(opt.last_engine_call_us)
        opt.time_engine = False
        eng_us = round(sorted(calls)[len(calls) // 2], 1)
        # the stand-in networks' own share (tests/standins.LookupDepth FINDS the stored map of every image by comparing images and reads the
        # winner back to the host: it costs more than the refinement; a real network's time would stand here instead)
        t_img, srcs_ = data[0]["color_left"], data[1]["color_left"]
        imgs = torch.cat([t_img] + list(srcs_) + [torch.flip(t_img, [3])], 0)
        nets = []
        for _ in range(5):
            torch.cuda.synchronize(); t1 = time.perf_counter()
            opt._disparities(imgs)
            torch.cuda.synchronize(); nets.append((time.perf_counter() - t1) * 1e6)
        net_us = round(sorted(nets)[2], 1)
        res[tag] = {"us_per_window": round(dt * 1e6, 1), "windows": n, "engine_call_us": eng_us, "standin_depth_net_us": net_us,
                    "shim_own_us": round(dt * 1e6 - eng_us - net_us, 1),
                    "over_engine_call": None if not eng_us else round(dt * 1e6 / eng_us, 2),
                    "over_engine_call_without_the_standin_depth_net": None if not eng_us else round((dt * 1e6 - net_us) / eng_us, 2)}
    res["what"] = (f"optimize_window(B={B}, S={S}, {W}x{H}, {ITER} PoseNet iterations, 4 GN iterations) with stand-in pose / depth networks (tests/standins.py: a lookup "
                   "depth net and a linear pose net, so the figure is the shim's own cost: tensor plumbing, the coupled pose initialisation, the engine call, "
                   "flip post-processing, D2H of the result dict); engine_call_us = the refine call inside it, device-synchronised, measured on its own")
    return res


def self_launch(n):
    """`python bench.py --gpus N` outside torch.distributed.run: this process (which has not touched the GPU and will not) starts
    N ranks of this file, one per GPU, and returns their exit code."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    return subprocess.call(cmd, env=env)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--warmup", type=int, default=200)
    ap.add_argument("--min-blocks", type=int, default=5, help="timed K-step blocks at least (the median block is reported; short blocks are "
                    "repeated until 50 ms anyway)")
    ap.add_argument("--windows-per-gpu", type=int, default=0, help="windows (B) per rank and step; default 1 on one GPU (config 2), 8 on several (config 3)")
    ap.add_argument("--total-windows", type=int, default=0, help="STRONG scaling instead of the default weak scaling: this many windows per step in "
                    "the whole job (BASELINE config 3: 64), split evenly over the ranks (SURVEY 8e: report both)")
    ap.add_argument("--sat-windows", type=int, default=32, help="windows per call of the chip-filling roofline leg (0 = skip)")
    ap.add_argument("--cpu-sample", type=float, default=20.0, help="seconds of wall time given to the CPU baseline (0 = skip)")
    ap.add_argument("--lanes", type=int, default=4, help="refine calls kept in flight (lanes of the handle, include/tcsfm.h); 1 = strictly one after the other")
    ap.add_argument("--ring-mb", type=float, default=320.0, help="the steps rotate over distinct calls whose inputs add up to at least this many MB "
                    "(> the 256 MiB Infinity Cache: a step's images come from HBM); 0 = every step re-runs ONE call (the round-3 protocol)")
    ap.add_argument("--coalesce", type=int, default=10, help="also time the steps as QUEUED calls that the library merges into one launch sequence per "
                    "this many calls (tcsfm_refine_window_queued, include/tcsfm.h; bit-identical per window); the faster of lanes / merged is the "
                    "headline, the other is reported beside it; 0 = lanes only")
    ap.add_argument("--coalesce-lanes", type=int, default=2, help="streams of the handle the merged sequences alternate over (tcsfm_set_coalesce_lanes): "
                    "the solve kernels of one sequence overlap the chip-filling launches of the other; 1 = the handle's stream only")
    ap.add_argument("--graph-replay", default="auto", choices=("auto", "0", "1"),
                    help="1: the handle replays the (repeated) refine call of every lane as one captured HIP graph (tcsfm_set_graph_replay: one "
                         "host launch per call instead of nine, same kernels, bit-identical results); 0: plain launches; auto: untimed blocks of "
                         "both before the timed region, the faster mode is timed (a slow host favours replay, a fast one plain launches)")
    ap.add_argument("--dump-poses", default="", help="rank 0 writes the gathered refined poses [world, pairs, 6] to this .npy file (tests)")
    ap.add_argument("--modes-budget", type=float, default=12.0, help="seconds given to the `modes` block (BASELINE configs 4 / 5, the KITTI window, the "
                    "reference-loss dense mode: windows/s and roofline fraction each); 0 = skip")
    ap.add_argument("--shim-sample", type=float, default=4.0, help="seconds given to the `shim` block (DepthOptimizer.optimize_window end to end); 0 = skip")
    args = ap.parse_args()

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:     # started the way the driver starts it: launch the ranks, touch no GPU here
        raise SystemExit(self_launch(args.gpus))

    import numpy as np
    # this application owns its process: it sets the two HIP runtime variables the launch structure likes BEFORE HIP initialises (the package
    # itself no longer touches the environment on import: tightly_coupled_sfm_amd/_lib.py)
    os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")
    os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: start N ranks with torch.distributed.run, or unset WORLD_SIZE and let bench.py launch them")
    distributed = world > 1 or bool(os.environ.get("TCSFM_BENCH_FORCE_DIST"))     # (forced: the RCCL path on ONE rank, tests/test_gpu_bench.py)
    # rehearsal switches (tests/test_gpu_bench.py): several ranks sharing ONE card over gloo -- RCCL refuses duplicate devices
    backend = os.environ.get("TCSFM_BENCH_BACKEND", "nccl")
    if os.environ.get("TCSFM_BENCH_ONE_DEVICE"):
        local_rank = 0
    torch.cuda.set_device(local_rank)
    coll_dev = "cuda" if backend == "nccl" else "cpu"
    coll_world = None
    if distributed:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)
        coll_world = dist.get_world_size()

    from tightly_coupled_sfm_amd import synth
    from tightly_coupled_sfm_amd.engine import Engine, default_opts

    B = args.windows_per_gpu or (8 if world > 1 else 1)
    if args.total_windows:
        assert args.total_windows % world == 0, "--total-windows must be a multiple of the number of ranks"
        B = args.total_windows // world
    npairs = 2 * SOURCES * B
    lanes = max(1, args.lanes)
    # rank r owns windows r*B .. r*B+B-1 of the global minibatch (contiguous block split, tightly_coupled_sfm_amd/parallel.py);
    # a window = the fwd + inv directed pair of one (target, source) frame pair.  Ring entry 0 is that minibatch; the further entries
    # are other minibatches of the same shape (other seeds), so that consecutive steps read different images.
    call_bytes = B * (2 * 3 + 2) * H * W * 4                 # target + source colours, two depth maps per window
    R = lanes if args.ring_mb <= 0 else int(-(-max(1.0, args.ring_mb * 1e6 / call_bytes) // lanes) * lanes)    # multiple of the lanes: a call
    R = max(R, lanes)                                                                                         # always runs on the same lane
    opts = default_opts(n_iters=ITERS)

    def make_call(j):
        # the window form of a batch (the library forms the fwd / inv pairs itself, bit-identical to the pair form): targets [B], the
        # S=1 source of each, initial poses in the stacked order [forward pairs | inverse pairs]
        b = synth.make_batch(npairs, H, W, seed0=100 * rank + 7919 * j, both_directions=True)
        dev = {k: torch.as_tensor(v).cuda().contiguous() for k, v in b.items()}
        win = dict(tgt=dev["tgt"][0::2].contiguous(), srcs=dev["src"][0::2].contiguous()[None], depth_t=dev["depth_t"][0::2].contiguous(),
                   depth_s=dev["depth_s"][0::2].contiguous()[None], K=dev["K"][0::2].contiguous(),
                   pose=torch.cat([dev["pose_init"][0::2], dev["pose_init"][1::2]]).contiguous())
        win["out"] = torch.empty_like(win["pose"])
        return (dev if j == 0 else None), win

    dev, win0 = make_call(0)
    ring = [win0] + [make_call(j)[1] for j in range(1, R)]
    for w in ring[1:]:       # ONE camera: every call passes the same intrinsics tensor, as the windows of a sequence do (the library
        assert torch.equal(w["K"], win0["K"])                # validates a device intrinsics pointer, with a blocking copy, the first
        w["K"] = win0["K"]                                   # time it sees it -- 84 different pointers would thrash its 4-entry cache)
    gt_w = torch.cat([dev["pose_gt"][0::2], dev["pose_gt"][1::2]])
    # The handle is created AFTER the inputs are on the card, as a pipeline that loads its data first would: on this ROCm the order in
    # which a process creates its streams decides how the lanes' hardware queues are placed, and a handle created before the process's
    # first device work can end up with lanes that slow each other down (4 lanes 10 000 instead of 24 000 frame-pairs/s, reproducibly:
    # scripts/lane_order_probe.py, profiles/r04_lane_order_probe.txt; DESIGN section 4 "Lanes")
    coal = 0 if args.coalesce <= 1 else max(1, min(args.coalesce, 16))               # queued calls per merged sequence (the library's table holds 16;
                                                                                     # the handle below is created for exactly that many calls' pairs)
    eng = Engine(H, W, npairs * max(1, coal), lanes=lanes)
    eng.use_own_stream()                     # lane 0 on the handle's own non-blocking stream, like the other lanes
    torch.cuda.synchronize()
    rot = [True]         # False: every step re-runs ring entries 0 .. lanes-1 (hot caches, the round-3 protocol)

    def step_k(k, nl):   # step k of a run with nl calls in flight: its ring entry on its lane; every step starts from its window's
        w = ring[k % R] if rot[0] else ring[k % nl]          # initial poses and writes the refined poses to the window's output
        if queued[0]:
            eng.refine_window_queued(w["tgt"], w["srcs"], w["depth_t"], w["depth_s"], w["K"], w["pose"], w["out"], opts)
        else:
            eng.refine_window_async(k % nl, w["tgt"], w["srcs"], w["depth_t"], w["depth_s"], w["K"], w["pose"], w["out"], opts)

    queued = [False]     # True: the steps are queued calls that the library merges (tcsfm_refine_window_queued) instead of lane calls

    def fence(nl):     # the contract's bracket: device-wide synchronise (it covers the lanes' streams) + barrier
        if queued[0]:
            eng.flush()                      # what is still waiting is launched inside the timed block
        torch.cuda.synchronize()
        if distributed:
            dist.barrier()

    enqueue_s = []       # host time spent enqueueing the K steps of a block (inside the block's wall time)
    kpos = [0]           # the ring position carries over from block to block

    def block(nl):
        fence(nl)
        k0 = kpos[0]
        t0 = time.perf_counter()
        for k in range(k0, k0 + args.steps):
            step_k(k, nl)
        t1 = time.perf_counter()
        fence(nl)
        kpos[0] = (k0 + args.steps) % (R * nl // np.gcd(R, nl))
        enqueue_s.append(t1 - t0)
        return time.perf_counter() - t0

    def timed(nl):
        """W warm-up steps, then R blocks of exactly K steps with `nl` calls in flight -> (median block seconds, sorted blocks)"""
        for k in range(args.warmup):
            step_k(k, nl)
        blocks = [block(nl)]
        reps = int(min(64, max(args.min_blocks, np.ceil(0.05 / max(blocks[0], 1e-9)))))
        if distributed:      # every rank runs the same number of blocks
            r = torch.tensor([reps], device=coll_dev, dtype=torch.int64)
            dist.all_reduce(r, op=dist.ReduceOp.MAX)
            reps = int(r.item())
        for _ in range(reps - 1):
            blocks.append(block(nl))
        bt = torch.tensor(blocks, device=coll_dev, dtype=torch.float64)
        if distributed:
            dist.all_reduce(bt, op=dist.ReduceOp.MAX)       # per block: the slowest rank
        blocks = sorted(bt.cpu().tolist())
        med = blocks[len(blocks) // 2] if len(blocks) % 2 else 0.5 * (blocks[len(blocks) // 2 - 1] + blocks[len(blocks) // 2])
        return med, blocks

    windows_per_block = args.steps * B * world
    graph_slots = min(64, max(4, R // lanes + 1))            # every (lane, ring entry) call is captured once

    def set_mode(replay):    # launch mode of the lanes' calls; replay: every call of the ring captured now, outside any timed block
        eng.set_graph_replay(graph_slots if replay else 0)
        if replay:
            for _ in range(3):
                for k in range(R):
                    step_k(k, lanes)
        fence(lanes)

    def probe(replay):       # untimed: median of a few K-step blocks in this mode
        set_mode(replay)
        for k in range(args.warmup):
            step_k(k, lanes)
        ts = sorted(block(lanes) for _ in range(int(min(16, max(3, np.ceil(0.01 * 1e3 / max(args.steps * 0.05, 1e-9)))))))
        return ts[len(ts) // 2]

    if args.graph_replay == "auto":
        t_mode = torch.tensor([probe(False), probe(True)], device=coll_dev, dtype=torch.float64)
        if distributed:
            dist.all_reduce(t_mode, op=dist.ReduceOp.MAX)     # every rank takes the same decision: the slowest rank's times
        use_replay = bool(t_mode[1] < t_mode[0])
    else:
        use_replay = args.graph_replay == "1"
    set_mode(use_replay)
    del enqueue_s[:]
    elapsed, blocks = timed(lanes)
    host_enqueue_us = float(np.median(enqueue_s)) / args.steps * 1e6
    for l in range(lanes):                   # a deferred device-side error of any lane surfaces here
        eng.lane_synchronize(l)
    # the round-3 protocol beside it: the same blocks re-running ONE window per lane (inputs, packs and records all cache resident)
    hot = None
    if R > lanes:
        rot[0] = False
        h_el, _ = timed(lanes)
        rot[0] = True
        for l in range(lanes):
            eng.lane_synchronize(l)
        hv, cv = windows_per_block / h_el, windows_per_block / elapsed
        hot = {"value": round(hv, 2), "ms_per_step": round(h_el / args.steps * 1e3, 5), "value_over_ring_value": round(hv / cv, 4),
               "what": f"the same K-step blocks re-running one window per lane (working set {lanes} x ~9 MB: Infinity-Cache resident) -- what rounds 1-3 "
                       f"reported; the headline rotates over {R} distinct calls ({R * call_bytes / 1e6:.0f} MB of inputs)"}
    # the same blocks in the OTHER launch mode, for comparison; bit-identical poses
    mine_out = [w["out"].clone() for w in ring]
    counts = eng.graph_replay_counts()
    set_mode(not use_replay)
    del enqueue_s[:]
    o_el, _ = timed(lanes)
    for l in range(lanes):
        eng.lane_synchronize(l)
    counts2 = eng.graph_replay_counts()
    other = {"mode": "plain launches (9 per call)" if use_replay else "graph replay", "value": round(windows_per_block / o_el, 2),
             "ms_per_step": round(o_el / args.steps * 1e3, 5), "host_enqueue_us_per_step": round(float(np.median(enqueue_s)) / args.steps * 1e6, 2),
             "same_poses": bool(all(torch.equal(a_, w["out"]) for a_, w in zip(mine_out, ring))),
             "captures": max(counts[0], counts2[0]), "replays": max(counts[1], counts2[1])}
    eng.set_graph_replay(0)
    # the same steps as QUEUED calls merged by the library into one launch sequence per `coal` calls (one stream: no reliance on how the
    # runtime places the lanes' hardware queues); per window the same bits
    merged = None
    if coal > 1:
        lanes_out = [w["out"].clone() for w in ring]
        eng.set_coalesce(coal)
        coal_streams = max(1, min(args.coalesce_lanes, lanes))
        eng.set_coalesce_lanes(coal_streams)
        queued[0] = True
        del enqueue_s[:]
        c_el, c_blocks = timed(lanes)
        queued[0] = False
        eng.set_coalesce_lanes(1)
        eng.set_coalesce(0)
        merged = {"calls_per_sequence": coal, "streams": coal_streams, "value": round(windows_per_block / c_el, 2), "ms_per_step": round(c_el / args.steps * 1e3, 5),
                  "host_enqueue_us_per_step": round(float(np.median(enqueue_s)) / args.steps * 1e6, 2),
                  "same_poses": bool(all(torch.equal(a_, w["out"]) for a_, w in zip(lanes_out, ring))),
                  "what": "tcsfm_refine_window_queued: the library runs every `calls_per_sequence` queued calls as ONE pack / (linearise, solve) x 4 "
                          "sequence over all their directed pairs (pointer table), the rest of a block at its closing flush"}
        lanes_res = {"calls_in_flight": lanes, "value": round(windows_per_block / elapsed, 2), "ms_per_step": round(elapsed / args.steps * 1e3, 5),
                     "mode": "graph replay" if use_replay else "plain launches (9 per call)"}
        if c_el < elapsed:          # the merged sequences are the faster way of running these steps: they are the headline
            elapsed, blocks = c_el, c_blocks
            host_enqueue_us = merged["host_enqueue_us_per_step"]
            merged["timed"] = True
        else:
            merged["timed"] = False
    # one call in flight: the host is not what binds (74 us of GPU time against 42 us of launches per call) and a graph launch adds
    # ~4 us of GPU time to the call -- the latency figure uses plain launches
    single = timed(1) if lanes > 1 else (elapsed, blocks)       # the same steps strictly one after the other
    eng.lane_synchronize(0)
    pose_io = ring[0]["out"]

    # final gather of the refined poses (RCCL over xGMI), outside the timed region; timed on its own (second call: no setup cost)
    final = pose_io.clone()
    gather_us = None
    if distributed:
        send = final.to(coll_dev)
        for k in range(2):
            gathered = [torch.empty_like(send) for _ in range(world)]
            fence(lanes)
            t0 = time.perf_counter()
            dist.all_gather(gathered, send)
            if coll_dev == "cuda":
                torch.cuda.synchronize()
            gather_us = (time.perf_counter() - t0) * 1e6
        final_all = torch.stack(gathered)
    else:
        final_all = final[None]
    assert torch.isfinite(final_all).all()
    if args.dump_poses and rank == 0:
        np.save(args.dump_poses, final_all.cpu().numpy())

    # instrumented pass on EVERY rank: in-kernel brackets + HIP events around every kernel launch, same steps (ring rotation, one call
    # in flight: the kernel has the chip -- what a roofline compares against)
    n_prof = min(max(args.steps, 300), 500)                 # enough launches for a stable average, whatever K the driver asked for
    # (in five chunks, each with its own totals: the average is over ALL launches as before; the per-chunk averages show the spread -- with ONE
    # call in flight the chip is lightly loaded and its shader clock moves between the 2.4 GHz and 1.4 GHz power states, which a VALU-bound
    # kernel's duration follows: 8.4-8.5 us per launch in the high state, 9.4-9.6 us when the governor has stepped down; an untimed 0.3 s
    # warm-up of the same load was tried and does not pin the state -- the light load itself is what lets the clock fall:
    # profiles/r04_bench_repeats.jsonl, scripts/diag/clock_sampler.sh)
    n_chunks = 5
    chunk_us, prof = [], {}
    for ci in range(n_chunks):
        eng.profile_begin()
        for k in range(ci * n_prof // n_chunks, (ci + 1) * n_prof // n_chunks):
            step_k(k, 1)
        pc = eng.profile_end()
        chunk_us.append(round(pc["linearize_kernel"][0] / max(pc["linearize_kernel"][1], 1) * 1e3, 3))
        for key, (ms_, n_) in pc.items():
            a_, b_ = prof.get(key, (0.0, 0))
            prof[key] = (a_ + ms_, b_ + n_)
    alg_bytes = 32 * H * W * npairs                          # 32 B/pixel/pair/iteration x pixels x pairs/launch
    k_ms, k_n = prof["linearize_kernel"]
    e_ms, e_n = prof["linearize"]
    avg_s = k_ms / max(k_n, 1) * 1e-3
    ev_s = e_ms / max(e_n, 1) * 1e-3
    mine = {"rank": rank, "avg_launch_us": round(avg_s * 1e6, 3), "achieved": round(alg_bytes / avg_s / 1e9, 2),
            "frac": round(alg_bytes / avg_s / 1e9 / HBM_PEAK_GBPS, 5)}
    # the same launches while `lanes` calls are in flight, as in the timed region: each k_linearize then shares the chip with the
    # other lane's k_solve / k_pack / k_linearize, so ITS duration is longer although the chip as a whole gets more done
    inflight = None
    if lanes > 1:
        eng.profile_begin()
        for k in range(n_prof):
            step_k(k, lanes)
        for l in range(lanes):
            eng.lane_synchronize(l)
        prof2 = eng.profile_end()
        k2_ms, k2_n = prof2["linearize_kernel"]
        a2 = k2_ms / max(k2_n, 1) * 1e-3
        inflight = {"steps_in_flight": lanes, "avg_launch_us": round(a2 * 1e6, 3), "achieved": round(alg_bytes / a2 / 1e9, 2),
                    "frac": round(alg_bytes / a2 / 1e9 / HBM_PEAK_GBPS, 5), "launches": int(k2_n),
                    "avg_launch_us_hip_events": round(prof2["linearize"][0] / max(prof2["linearize"][1], 1) * 1e3, 3)}
    # ... and the launches of the TIMED mode when that is the merged sequences: one k_linearize launch then covers the directed pairs of
    # `coal` queued calls (tcsfm_refine_window_queued); bracketed in-kernel like the others (profiling does not stop the merging)
    timed_mode = None
    if merged and merged["timed"]:
        eng.set_coalesce(coal)
        eng.set_coalesce_lanes(max(1, min(args.coalesce_lanes, lanes)))
        queued[0] = True
        eng.profile_begin()
        for k in range(coal * 30):
            step_k(k, lanes)
        eng.flush()
        for l in range(lanes):
            eng.lane_synchronize(l)
        prm = eng.profile_end()
        queued[0] = False
        eng.set_coalesce_lanes(1)
        eng.set_coalesce(0)
        km_ms, km_n = prm["linearize_kernel"]
        am = km_ms / max(km_n, 1) * 1e-3
        alg_m = alg_bytes * coal
        timed_mode = {"mode": f"{coal} queued B={B} calls per launch sequence, sequences alternating over {max(1, min(args.coalesce_lanes, lanes))} streams",
                      "pairs_per_launch": npairs * coal, "avg_launch_us": round(am * 1e6, 3), "launches": int(km_n), "algorithmic_bytes_per_launch": alg_m,
                      "achieved": round(alg_m / am / 1e9, 2), "frac": round(alg_m / am / 1e9 / HBM_PEAK_GBPS, 5), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                      "frac_source": "this run: in-kernel s_memrealtime bracket of the merged k_linearize launches, as they run in the timed region "
                                     "(two sequences in flight on two streams: a launch shares the chip with the other sequence's kernels)"}
        bm_ms, bm_n = prm.get("linearize_busy", (0.0, 0))
        if bm_ms > 0 and bm_n == km_n:
            # the launches of the two streams overlap: the chip's rate on the kernel = all their bytes / the time at least one of them ran
            timed_mode["chip_level"] = {"busy_us": round(bm_ms * 1e3, 1), "launch_us_summed": round(km_ms * 1e3, 1), "overlap": round(km_ms / bm_ms, 3),
                                        "achieved": round(alg_m * km_n / (bm_ms * 1e-3) / 1e9, 2), "frac": round(alg_m * km_n / (bm_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS, 5),
                                        "what": "algorithmic bytes of ALL merged k_linearize launches of the pass / the time during which at least one of them was "
                                                "running (union of the in-kernel [start, end] brackets over both streams): what the chip sustains on the kernel in the timed mode"}
    per_rank = [mine]
    if distributed:
        per_rank = [None] * world
        dist.all_gather_object(per_rank, mine)
    roof = None
    if rank == 0:
        rp = rocprof_avg_us() if B == 1 else None
        meta = profiles_meta()
        # Headline: THIS run's in-kernel bracket.  The committed rocprofv3 average of the same kernel stands beside it and must agree
        # (it includes ~1.2 us of dispatch and completion): frac_consistent.
        committed = None
        if rp and "lanes_1" in rp:
            c_us = rp["lanes_1"]["us"]
            committed = {"avg_launch_us": c_us, "achieved": round(alg_bytes / (c_us * 1e-6) / 1e9, 2),
                         "frac": round(alg_bytes / (c_us * 1e-6) / 1e9 / HBM_PEAK_GBPS, 5), "file": rp["lanes_1"]["file"],
                         "what": "rocprofv3 --kernel-trace --stats AverageNs of the committed run of `python bench.py --lanes 1`",
                         "dispatch_offset_us": DISPATCH_OFFSET_US}
        traffic = load_pmc() if B == 1 else None
        call_traffic = load_pmc("hbm_bytes_per_call") if B == 1 else None
        vb = valu_bound(avg_s, npairs)
        roof = {"bound": "hbm", "bound_actual": "valu", "achieved": mine["achieved"], "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": mine["frac"],
                "frac_source": "this run: in-kernel s_memrealtime bracket of every k_linearize launch (earliest workgroup start -> latest workgroup "
                               "end), one call in flight, steps rotating over the ring",
                "avg_launch_us": mine["avg_launch_us"], "avg_launch_us_hip_events": round(ev_s * 1e6, 3),
                "avg_launch_us_chunks": chunk_us,
                "valu_frac_of_bound": None if vb is None else vb["frac_of_bound"],
                "traffic": traffic, "traffic_over_algorithmic": None if traffic is None else round(traffic / alg_bytes, 3),
                "traffic_note": "below 1 since round 5: the algorithmic figure (32 B / pixel / directed pair: a target texel + the source taps) counts the two "
                                "image packs of a window once per directed pair, and the window forms now pack every image ONCE -- a pair reads its target "
                                "from its partner's source pack (LinParams::tshare), so the two pairs of a window fetch the same lines",
                "kernel": "k_linearize", "launches": int(k_n), "algorithmic_bytes_per_launch": alg_bytes,
                "rocprof_committed": committed,
                "frac_consistent": None if committed is None else bool(consistent(mine["avg_launch_us"], committed["avg_launch_us"])),
                "profiles_match_source": None if meta is None else meta["matches"], "profiles_meta": meta,
                "rocprof_avg_us": rp,
                "rocprof_note": "profiles/<tag>_kernel_stats.csv: rocprofv3 --kernel-trace --stats of this very command (its average covers the "
                                "launches of the timed blocks with `steps_in_flight` calls in flight, of the single-stream blocks and of both instrumented passes); "
                                "profiles/<tag>_lanes1_kernel_stats.csv: the same command with --lanes 1 (every launch has the chip); "
                                "profiles/<tag>_sat_kernel_stats.csv: scripts/sat_workload.py (32 windows per call)",
                "other_kernels_avg_us_hip_events": {k: round(v[0] / max(v[1], 1) * 1e3, 3) for k, v in prof.items() if k in ("solve", "pack")},
                "valu_bound": vb,
                "whole_call": {"algorithmic_bytes": ITERS * alg_bytes, "traffic_bytes": call_traffic,
                               "traffic_over_algorithmic": None if call_traffic is None else round(call_traffic / (ITERS * alg_bytes), 3),
                               "what": "all launches of one call (pack + 4 x (linearise + solve)) against 4 x the algorithmic bytes of a linearisation"},
                "measured_with": "ONE call in flight (the kernel has the chip: what a roofline compares against); `in_flight` repeats the live bracket with the "
                                 "lanes' calls in flight, `timed_mode` brackets the launches of the mode `value` was timed in",
                "in_flight": inflight, "timed_mode": timed_mode}
        if distributed:
            roof["per_rank"] = per_rank

    # the same kernel with the chip full (32 windows = 64 directed pairs per call): the B=1 figure above is bounded by
    # launch latency and a grid of only 480 workgroups, this one by the kernel itself (report both)
    roof_sat = None
    if rank == 0 and world == 1 and args.sat_windows > 0:
        rep = args.sat_windows
        two = {k: dev[k][:2] for k in ("tgt", "src", "depth_t", "depth_s", "K", "pose_init")}
        big = {k: two[k].repeat((rep,) + (1,) * (two[k].dim() - 1)).contiguous() for k in two}
        eng_b = Engine(H, W, 2 * rep)
        out_b = torch.empty_like(big["pose_init"])
        run_b = lambda: eng_b.refine_into(big["tgt"], big["src"], big["depth_t"], big["depth_s"], big["K"], big["pose_init"], out_b, opts)
        for _ in range(5):
            run_b()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(30):
            run_b()
        torch.cuda.synchronize()
        wall = (time.perf_counter() - t0) / 30
        eng_b.profile_begin()
        for _ in range(30):
            run_b()
        prof_b = eng_b.profile_end()
        kb_ms, kb_n = prof_b["linearize_kernel"]
        avg_b = kb_ms / max(kb_n, 1) * 1e-3
        alg_b = 32 * H * W * 2 * rep
        rps = (rocprof_avg_us() or {}).get("saturated") if rep == 32 else None
        vbs = valu_bound(avg_b, 2 * rep)
        roof_sat = {"workload": f"{rep} windows ({2 * rep} directed pairs) per call", "achieved": round(alg_b / avg_b / 1e9, 2),
                    "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": round(alg_b / avg_b / 1e9 / HBM_PEAK_GBPS, 5),
                    "frac_source": "this run: in-kernel s_memrealtime bracket", "avg_launch_us": round(avg_b * 1e6, 2),
                    "avg_launch_us_hip_events": round(prof_b["linearize"][0] / max(prof_b["linearize"][1], 1) * 1e3, 2),
                    "rocprof_committed": None if not rps else {"avg_launch_us": rps["us"], "frac": round(alg_b / (rps["us"] * 1e-6) / 1e9 / HBM_PEAK_GBPS, 5),
                                                               "file": rps["file"]},
                    "frac_consistent": None if not rps else bool(abs(avg_b * 1e6 - rps["us"]) <= 0.15 * rps["us"]),
                    "bound_actual": "valu", "valu_frac_of_bound": None if vbs is None else vbs["frac_of_bound"],
                    "algorithmic_bytes_per_launch": alg_b, "frame_pairs_per_s": round(rep / wall, 1), "valu_bound": vbs}
        del eng_b, big, out_b

    if rank == 0:
        windows_per_block = args.steps * B * world
        # sanity figures from one extra, untimed call: the cost the 4 linearisations saw, and how far the refined poses are from
        # the scene's true poses
        _, _, st = eng.refine(dev["tgt"], dev["src"], dev["depth_t"], dev["depth_s"], dev["K"], dev["pose_init"], opts, stats=True)
        eng.synchronize()           # (the handle runs on its own stream: its outputs are read by torch only after this)
        cost_traj = [round(float(x), 6) for x in st[:, :ITERS, 0].mean(0).cpu()]
        err_t = float((final[:, :3] - gt_w[:, :3]).norm(dim=1).mean() / gt_w[:, :3].norm(dim=1).mean())
        err_0 = float((ring[0]["pose"][:, :3] - gt_w[:, :3]).norm(dim=1).mean() / gt_w[:, :3].norm(dim=1).mean())
        # The timed windows are rendered with a plain pinhole camera, while the reference's sampler (stn.py:198-231,266) is offset by up to half a
        # pixel of flow from it: the minimiser of the reference's residual on them is NOT the scene truth (hence ~4 % above).  The same scene
        # rendered THROUGH the reference's sampling model (synth.make_pair(sampler_consistent=True): only the sampled role can be made
        # consistent, so forward pairs only) has the truth as its minimiser: one untimed call each with this run's 4 Gauss-Newton iterations
        # and with 16 Levenberg-Marquardt iterations
        tb = synth.make_batch(2, H, W, seed0=4242, sampler_consistent=True)
        td = {k: torch.as_tensor(v).cuda().contiguous() for k, v in tb.items()}
        def _truth(o_):
            p_ = eng.refine(td["tgt"], td["src"], td["depth_t"], td["depth_s"], td["K"], td["pose_init"], o_)[0]
            eng.synchronize()
            g_ = td["pose_gt"]
            return {"translation_rel": round(float((p_[:, :3] - g_[:, :3]).norm(dim=1).mean() / g_[:, :3].norm(dim=1).mean()), 5),
                    "rotation_deg": round(float(torch.rad2deg((p_[:, 3:] - g_[:, 3:]).norm(dim=1)).mean()), 4)}
        from tightly_coupled_sfm_amd import _lib as _L
        g0_ = td["pose_gt"]; i0_ = td["pose_init"]
        truth = {"what": "two forward pairs of the same kind of scene rendered through the reference's own sampling model (the residual's minimiser is "
                         "the scene truth there), perturbed as the timed windows are; untimed",
                 "initial": {"translation_rel": round(float((i0_[:, :3] - g0_[:, :3]).norm(dim=1).mean() / g0_[:, :3].norm(dim=1).mean()), 5),
                             "rotation_deg": round(float(torch.rad2deg((i0_[:, 3:] - g0_[:, 3:]).norm(dim=1)).mean()), 4)},
                 "gn_4_iterations": _truth(default_opts(n_iters=ITERS)),
                 "lm_16_iterations": _truth(default_opts(n_iters=16, solver=_L.SOLVER_LM))}
        cfg_name = ("KITTI-like 640x192, batch=1 frame-pair (fwd+inv directed pairs), 4 GN iters, 6-DoF pose" if B == 1 else
                    f"KITTI-like 640x192, {B * world} frame-pairs sharded over {world} GPU(s) ({B} windows = {npairs} directed pairs per GPU and step), 4 GN iters, 6-DoF pose")
        # how the timed steps were EXECUTED is part of the workload's name (every step is one B-window call; the calls are independent windows)
        if merged and merged["timed"]:
            timed_as = (f"{coal} queued B={B} calls per launch sequence (tcsfm_refine_window_queued: the library merges them into one pack / (linearise, solve) x 4 sequence "
                        f"over {npairs * coal} directed pairs; bit-identical per window); the B={B} call on its own: `single_stream`")
        else:
            timed_as = f"{lanes} independent B={B} calls in flight on the handle's lanes; the B={B} call on its own: `single_stream`"
        cfg_name += "; timed as " + timed_as
        out = {
            "metric": "optimized frame-pairs/sec at 640x192, 4 GN iters",
            "value": round(windows_per_block / elapsed, 2),
            "unit": "frame-pairs/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 5),
            "higher_is_better": True, "scaling": "strong" if args.total_windows else "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": cfg_name, "timed_as": timed_as, "lane_probe": eng.lane_probe() if lanes > 1 else None, "windows_per_gpu": B, "sources": SOURCES, "directed_pairs_per_step": npairs,
                       "global_batch_frame_pairs": B * world, "gn_iters": ITERS, "solver": "gn", "param": "se3",
                       "steps_in_flight": (coal if (merged and merged["timed"]) else lanes), "lanes": lanes, "ring_calls": R, "ring_input_MB": round(R * call_bytes / 1e6, 1),
                       "collective_backend": backend if distributed else None, "collective_world_size": coll_world,
                       "parallelism": f"{world} independent shards, no data-path collective; one all_gather of the poses after the timed region"},
            "timed_blocks": len(blocks),
            "ms_per_step_blocks": {"min": round(blocks[0] / args.steps * 1e3, 5), "median": round(elapsed / args.steps * 1e3, 5),
                                   "max": round(blocks[-1] / args.steps * 1e3, 5)},
            "hot_cache": hot,
            "single_stream": {"value": round(windows_per_block / single[0], 2), "ms_per_step": round(single[0] / args.steps * 1e3, 5),
                              "what": "the same K-step blocks with ONE call in flight (steps_in_flight = 1), plain launches: the per-call latency figure; "
                                      "the headline keeps `steps_in_flight` independent calls in flight on the handle's lanes, which fills "
                                      "the idle time between the short kernels of a B=1 call"},
            "host_enqueue_us_per_step": round(host_enqueue_us, 2),
            "launch_mode": {"timed": ("merged sequences: the steps are queued calls, the library runs them %d at a time as one launch sequence "
                                      "(tcsfm_refine_window_queued); bit-identical poses" % coal) if (merged and merged["timed"]) else
                                     ("graph replay: every call of the ring (same buffers each time round) is captured once and launched as ONE HIP graph "
                                      "(tcsfm_set_graph_replay); same kernels, bit-identical poses" if use_replay else "plain launches (9 per call)"),
                            "merged": merged, "lanes": None if not merged else lanes_res,
                            "chosen_by": ("untimed probe blocks of both modes before the timed region" if args.graph_replay == "auto" else "--graph-replay " + args.graph_replay),
                            "other_mode": other},
            "final_gather_us": None if gather_us is None else round(gather_us, 1),
            "roofline": roof,
            "roofline_saturated": roof_sat,
            "cpu_baseline": cpu_baseline(args.cpu_sample) if (args.cpu_sample > 0 and world == 1) else None,
            "modes": modes_block(args.modes_budget) if (args.modes_budget > 0 and world == 1) else None,
            "shim": shim_block(args.shim_sample) if (args.shim_sample > 0 and world == 1) else None,
            "check": {"mean_cost_at_each_linearisation": cost_traj,
                      "rel_translation_distance_to_scene_truth": {"initial": round(err_0, 5), "refined": round(err_t, 5),
                                                                   "note": "pinhole-rendered windows: the reference's sampler is offset from the renderer, see truth_sampler_consistent"},
                      "truth_sampler_consistent": truth},
        }
        print(json.dumps(out), flush=True)
    if distributed:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
