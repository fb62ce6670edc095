#!/usr/bin/env python3
"""Does the order in which a process creates its HIP streams change how well the handle's lanes overlap?  (It does, reproducibly.)

    python scripts/lane_order_probe.py            # one fresh process per case; prints frame-pairs/s with 1 and with 4 lanes

Cases: `engine_first` -- the handle (its lanes' streams) is the process's first device work; `touch_first` -- torch launches one tiny
kernel before; `data_first` -- torch uploads the inputs before the handle exists (what bench.py and the examples do);
`engine_first_burn` -- engine first, but one throw-away stream is created AND used before every lane's stream.
Measured round 4 (profiles/r04_lane_order_probe.txt): engine_first 4 lanes ~10 000 frame-pairs/s (SLOWER than one lane, 13 500);
data_first 23 000-26 000; a memory pad between the lanes' scratch changes nothing, stream priorities change nothing: it is the
placement of the streams' hardware queues (kernels of lanes on 'bad' queue combinations take 40-55 us each instead of 11 when two
of them overlap, rocprofv3 kernel trace), not the memory layout.  The mechanism inside ROCclr / the firmware scheduler was not
identified; the library creates plain non-blocking streams and the documentation tells callers to create the handle after their
first device work."""
import os, subprocess, sys, time
os.environ.setdefault("TCSFM_SET_ENV_DEFAULTS", "1")      # (a measurement script owns its process: HIP_FORCE_DEV_KERNARG / GPU_MAX_HW_QUEUES when absent)
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CASES = ("engine_first", "touch_first", "data_first", "engine_first_burn")


def run(case):
    sys.path.insert(0, ROOT)
    from tightly_coupled_sfm_amd import _lib  # noqa
    import torch
    from tightly_coupled_sfm_amd import synth
    from tightly_coupled_sfm_amd.engine import Engine, default_opts
    H, W, L = 192, 640, 4
    b = synth.make_batch(2, H, W, seed0=0, both_directions=True)
    eng = None
    if case == "engine_first":
        eng = Engine(H, W, 2, lanes=L)
    elif case == "touch_first":
        torch.zeros(1, device="cuda"); torch.cuda.synchronize()
        eng = Engine(H, W, 2, lanes=L)
    elif case == "engine_first_burn":
        keep = []

        def burn():
            s = torch.cuda.Stream()
            with torch.cuda.stream(s):
                keep.append(torch.zeros(64, device="cuda"))
            s.synchronize(); keep.append(s)
        burn(); eng = Engine(H, W, 2, lanes=1)
        for n in range(2, L + 1):
            burn(); eng.set_lanes(n)
    dev = {k: torch.as_tensor(v).cuda().contiguous() for k, v in b.items()}
    if eng is None:
        eng = Engine(H, W, 2, lanes=L)
    eng.use_own_stream()
    win = dict(tgt=dev["tgt"][0::2].contiguous(), srcs=dev["src"][0::2].contiguous()[None], depth_t=dev["depth_t"][0::2].contiguous(),
               depth_s=dev["depth_s"][0::2].contiguous()[None], K=dev["K"][0::2].contiguous(),
               pose=torch.cat([dev["pose_init"][0::2], dev["pose_init"][1::2]]).contiguous())
    outs = [torch.empty_like(win["pose"]) for _ in range(L)]
    o = default_opts(n_iters=4)
    torch.cuda.synchronize()
    res = {}
    nit = int(os.environ.get("PROBE_ITERS", "2000"))
    for nl in (1, L):
        for k in range(200):
            eng.refine_window_async(k % nl, win["tgt"], win["srcs"], win["depth_t"], win["depth_s"], win["K"], win["pose"], outs[k % nl], o)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for k in range(nit):
            eng.refine_window_async(k % nl, win["tgt"], win["srcs"], win["depth_t"], win["depth_s"], win["K"], win["pose"], outs[k % nl], o)
        torch.cuda.synchronize()
        res[nl] = nit / (time.perf_counter() - t0)
    print(f"{case:24s} 1 lane {res[1]:8.0f}   {L} lanes {res[L]:8.0f} frame-pairs/s", flush=True)


if __name__ == "__main__":
    if len(sys.argv) > 1:
        run(sys.argv[1])
    else:
        for c in CASES + CASES:
            subprocess.call([sys.executable, os.path.abspath(__file__), c])
