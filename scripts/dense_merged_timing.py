import sys, time, json
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import numpy as np, torch
import os as _os; _os.environ.setdefault("TCSFM_SET_ENV_DEFAULTS", "1")      # (a measurement script owns its process: HIP_FORCE_DEV_KERNARG / GPU_MAX_HW_QUEUES when absent)
from tightly_coupled_sfm_amd import synth
from tightly_coupled_sfm_amd.engine import Engine, default_opts
for H, W in ((240, 320), (256, 448)):
    ws = []
    for i in range(12):
        b = synth.make_batch(2, H, W, seed0=17 * i, both_directions=True)
        d = {k: torch.as_tensor(v).cuda().contiguous() for k, v in b.items()}
        ws.append(dict(tgt=d["tgt"][:1].contiguous(), srcs=d["src"][:1][None].contiguous(), dt=d["depth_t"][:1].contiguous(), ds=d["depth_s"][:1][None].contiguous(),
                       pose=torch.stack([d["pose_init"][0], d["pose_init"][1]]).contiguous(), po=torch.empty(2, 6, device="cuda"), do=torch.empty(2, 1, H, W, device="cuda")))
    K = torch.as_tensor(synth.make_batch(1, H, W)["K"][:1]).cuda().contiguous()
    o = default_opts(n_iters=4, min_depth=0.03, max_depth=3.0)
    torch.cuda.synchronize()
    e = Engine(H, W, 2 * 10, lanes=2)
    def single(n=300):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for k in range(n):
            w = ws[k % 12]; e.refine_dense_window(w["tgt"], w["srcs"], w["dt"], w["ds"], K, w["pose"], o)
        torch.cuda.synchronize(); return n / (time.perf_counter() - t0)
    def merged(n=600):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for k in range(n):
            w = ws[k % 12]; e.refine_dense_window_queued(w["tgt"], w["srcs"], w["dt"], w["ds"], K, w["pose"], w["po"], w["do"], o)
        e.flush(); torch.cuda.synchronize(); return n / (time.perf_counter() - t0)
    single(50); r1 = single()
    res = {}
    for calls, streams in ((10, 1), (10, 2), (6, 2)):
        e.set_coalesce(calls); e.set_coalesce_lanes(streams); merged(60); res[f"{calls}x{streams}"] = round(merged())
    e.set_coalesce_lanes(1); e.set_coalesce(0)
    print(json.dumps({"HxW": f"{H}x{W}", "dense B=1 windows/s, one call at a time": round(r1), "queued and merged (calls per sequence x streams)": res}))
    e.close()
