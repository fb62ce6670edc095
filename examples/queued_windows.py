#!/usr/bin/env python3
"""Many independent B = 1 windows, device-resident: the three ways of running them through one handle, with their rates.

    python examples/queued_windows.py [--windows 200]          (needs an MI355X)

  one call at a time        tcsfm_refine_window, nine dependent launches per window, the chip idles between them
  four calls in flight      tcsfm_refine_window_async on the handle's lanes (own stream and scratch per lane), every repeated call
                            replayed as one captured HIP graph (the host is the bottleneck otherwise)
  queued, merged by the library   tcsfm_refine_window_queued + tcsfm_set_coalesce(10) + tcsfm_set_coalesce_lanes(2): every ten queued calls
                            run as ONE pack / (linearise, solve) x n_iters sequence over all their directed pairs (a pointer table in the
                            kernel arguments reaches every call's own buffers), consecutive sequences alternate over two streams; tcsfm_flush
                            launches what is left and orders the handle's stream behind all of them.  Per window the poses are the same bits.
The handle is created AFTER the inputs are on the card (DESIGN.md section 4 "Lanes": how well a process's streams overlap depends on when it
created them)."""
import argparse
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import os as _os; _os.environ.setdefault("TCSFM_SET_ENV_DEFAULTS", "1")      # (a measurement script owns its process: HIP_FORCE_DEV_KERNARG / GPU_MAX_HW_QUEUES when absent)
from tightly_coupled_sfm_amd import synth                                    # noqa: E402
from tightly_coupled_sfm_amd.engine import Engine, default_opts              # noqa: E402

H, W = 192, 640


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--windows", type=int, default=200)
    ap.add_argument("--distinct", type=int, default=12, help="distinct synthetic windows (the run cycles over them)")
    args = ap.parse_args()
    t = lambda a: torch.as_tensor(np.ascontiguousarray(a, dtype=np.float32)).cuda()
    ws = []
    for i in range(args.distinct):
        p = synth.make_pair(H, W, seed=300 + i)
        q = synth.perturb_pose(p["pose_gt"], 300 + i)
        ws.append(dict(tgt=t(p["tgt"][None]), srcs=t(p["src"][None, None]), dt=t(p["depth_t"][None, None]), ds=t(p["depth_s"][None, None, None]),
                       pose=t(np.stack([q, -q])), out=torch.empty(2, 6, device="cuda")))
    K = t(synth.make_pair(H, W, seed=300)["K"][None])
    torch.cuda.synchronize()
    eng = Engine(H, W, 2 * 10, lanes=4)          # room for ten B = 1 calls per merged sequence
    o = default_opts(n_iters=4)
    n = args.windows

    def run(kind):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for k in range(n):
            w = ws[k % len(ws)]
            if kind == "single":
                eng.refine_window_async(0, w["tgt"], w["srcs"], w["dt"], w["ds"], K, w["pose"], w["out"], o)
            elif kind == "lanes":
                eng.refine_window_async(k % 4, w["tgt"], w["srcs"], w["dt"], w["ds"], K, w["pose"], w["out"], o)
            else:
                eng.refine_window_queued(w["tgt"], w["srcs"], w["dt"], w["ds"], K, w["pose"], w["out"], o)
        if kind == "queued":
            eng.flush()
        torch.cuda.synchronize()
        return n / (time.perf_counter() - t0), [w["out"].clone() for w in ws]

    run("single")                                 # warm-up
    r1, p1 = run("single")
    # nine launches + three event calls per lane call cost the host ~60 us from Python -- more than the GPU needs per call with four in
    # flight -- so the lanes replay every (lane, window) call as ONE captured HIP graph (tcsfm_set_graph_replay; same kernels, same bits)
    eng.set_graph_replay(max(4, (len(ws) + 3) // 4 + 1))
    run("lanes")                                  # (captures)
    run("lanes")
    r4, p4 = run("lanes")
    eng.set_graph_replay(0)
    eng.set_coalesce(10); eng.set_coalesce_lanes(2)
    run("queued")
    rq, pq = run("queued")
    eng.set_coalesce_lanes(1); eng.set_coalesce(0)
    same = all(torch.equal(a, b) and torch.equal(a, c) for a, b, c in zip(p1, p4, pq))
    print(f"one call at a time {r1:8.0f} windows/s | four calls in flight {r4:8.0f} | queued, merged ten at a time on two streams {rq:8.0f} | same poses: {same}")
    if r4 < 1.3 * r1:
        print("(the lanes of this process overlap badly -- how a process's HIP streams share the hardware queues depends on its history of stream "
              "creation, DESIGN.md section 4 'Lanes': the same four lanes reach 22-25 k windows/s in bench.py; the merged sequences do not depend on it)")
    assert same


if __name__ == "__main__":
    main()
