"""(the other tree: `mkdir ab_r02 && git archive <commit> tightly_coupled_sfm_amd include | tar -x -C ab_r02`, build it there with the flags of
build.py; ab_*/ is git-ignored but travels to the GPU box)
A/B of tcsfm_refine_sequence between library trees on ONE box: python scripts/experiments/seq_ab.py <tree-root>   (every run's windows/s)"""
import json, os, sys, time
root = os.path.abspath(sys.argv[1])
sys.path.insert(0, root)
import numpy as np, torch
import os as _os; _os.environ.setdefault("TCSFM_SET_ENV_DEFAULTS", "1")      # (a measurement script owns its process: HIP_FORCE_DEV_KERNARG / GPU_MAX_HW_QUEUES when absent)
from tightly_coupled_sfm_amd import synth
from tightly_coupled_sfm_amd.engine import Engine, default_opts
H, W, T = 192, 640, 200
seq = synth.make_sequence(T, H, W, seed=5)
frames = torch.as_tensor(seq["frames"]).pin_memory(); depths = torch.as_tensor(seq["depths"]).pin_memory()
K, init = seq["K"], torch.as_tensor(seq["init"])
opts = default_opts(n_iters=4)
o2 = default_opts(n_iters=4, argmin=1, w_dc=0.15)
step = seq["init"][:, 0]
init2 = torch.as_tensor(np.stack([np.stack([-step[w], step[w + 1], step[w], -step[w + 1]]) for w in range(T - 2)]).astype(np.float32))
for S, lanes, wpc in ((1, 1, 8), (1, 2, 8), (1, 2, 16), (1, 1, 16), (2, 2, 8)):
    e = Engine(H, W, 2 * S * wpc, lanes=lanes)
    kw = dict(windows_per_call=wpc) if S == 1 else dict(sources=2, windows_per_call=wpc, target_pos=-1)
    p0, o = (init, opts) if S == 1 else (init2, o2)
    e.refine_sequence(frames[:60], depths[:60], K, p0[:60 - S], o, **kw)
    ts = []
    for _ in range(9):
        t0 = time.perf_counter(); e.refine_sequence(frames, depths, K, p0, o, **kw); ts.append(time.perf_counter() - t0)
    print(json.dumps({"tree": os.path.basename(root), "S": S, "lanes": lanes, "wpc": wpc, "windows_per_s": [round((T - S) / t) for t in ts]}), flush=True)
    e.close()
