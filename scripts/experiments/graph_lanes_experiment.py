"""Throughput of B=1 calls when every call is ONE graph launch: G engines (one HIP stream each) replay a captured refine() round
robin.  Compares with the library's lanes (plain launches, host-bound at 9 launches per call)."""
import json, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import os as _os; _os.environ.setdefault("TCSFM_SET_ENV_DEFAULTS", "1")      # (a measurement script owns its process: HIP_FORCE_DEV_KERNARG / GPU_MAX_HW_QUEUES when absent)
from tightly_coupled_sfm_amd import synth
from tightly_coupled_sfm_amd.engine import Engine, default_opts
H, W, N = 192, 640, 3000
b = synth.make_batch(2, H, W, seed0=0, both_directions=True)
dev = {k: torch.as_tensor(v).cuda().contiguous() for k, v in b.items()}
o = default_opts(n_iters=4)
e0 = Engine(H, W, 2)
ref = torch.empty_like(dev["pose_init"])
e0.refine_into(dev["tgt"], dev["src"], dev["depth_t"], dev["depth_s"], dev["K"], dev["pose_init"], ref, o)
torch.cuda.synchronize()
for G in (1, 2, 3, 4):
    engines, streams, graphs, outs = [], [], [], []
    for g in range(G):
        e = Engine(H, W, 2)
        s = torch.cuda.Stream()
        out = torch.empty_like(dev["pose_init"])
        with torch.cuda.stream(s):
            e.use_torch_stream()
            for _ in range(3):
                e.refine_into(dev["tgt"], dev["src"], dev["depth_t"], dev["depth_s"], dev["K"], dev["pose_init"], out, o)
        torch.cuda.synchronize()
        cg = torch.cuda.CUDAGraph()
        with torch.cuda.graph(cg, stream=s):
            e.refine_into(dev["tgt"], dev["src"], dev["depth_t"], dev["depth_s"], dev["K"], dev["pose_init"], out, o)
        engines.append(e); streams.append(s); graphs.append(cg); outs.append(out)
    def sweep(n):
        for i in range(n):
            with torch.cuda.stream(streams[i % G]):
                graphs[i % G].replay()
    sweep(200); torch.cuda.synchronize()
    ts = []
    for _ in range(5):
        t0 = time.perf_counter(); sweep(N); t_host = time.perf_counter() - t0; torch.cuda.synchronize(); ts.append((time.perf_counter() - t0, t_host))
    ts.sort()
    t, th = ts[2]
    print(json.dumps({"graphs_in_flight": G, "windows_per_s": round(N / t, 1), "us_per_window": round(t / N * 1e6, 2),
                      "host_us_per_window": round(th / N * 1e6, 2), "same_result": all(torch.equal(x, ref) for x in outs)}), flush=True)
    del graphs, engines
