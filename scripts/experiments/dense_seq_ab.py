"""(the other tree: `mkdir ab_r02 && git archive <commit> tightly_coupled_sfm_amd include | tar -x -C ab_r02`, build it there with the flags of
build.py; ab_*/ is git-ignored but travels to the GPU box)
A/B of tcsfm_refine_dense_sequence between library trees on ONE box: python scripts/experiments/dense_seq_ab.py <tree-root>
(the tree's own package is imported; prints windows/s for lanes x windows_per_call)"""
import json, os, sys, time
root = os.path.abspath(sys.argv[1])
sys.path.insert(0, root)
import torch
import os as _os; _os.environ.setdefault("TCSFM_SET_ENV_DEFAULTS", "1")      # (a measurement script owns its process: HIP_FORCE_DEV_KERNARG / GPU_MAX_HW_QUEUES when absent)
from tightly_coupled_sfm_amd import synth
from tightly_coupled_sfm_amd.engine import Engine, default_opts

def run_sequence(H, W, lanes, wpc, T=120):
    seq = synth.make_sequence(T, H, W, seed=3)
    frames, depths = torch.as_tensor(seq["frames"]).pin_memory(), torch.as_tensor(seq["depths"]).pin_memory()
    init = torch.as_tensor(seq["init"])
    e = Engine(H, W, 2 * wpc, lanes=lanes)
    o = default_opts(n_iters=4, min_depth=0.03, max_depth=3.0)
    e.refine_dense_sequence(frames[:40], depths[:40], seq["K"], init[:39], o, windows_per_call=wpc)
    ts = []
    for _ in range(7):
        t0 = time.perf_counter(); e.refine_dense_sequence(frames, depths, seq["K"], init, o, windows_per_call=wpc); ts.append(time.perf_counter() - t0)
    print(json.dumps({"tree": root, "lanes": lanes, "wpc": wpc, "windows_per_s": [round((T - 1) / t) for t in ts]}), flush=True)
    e.close()

for lanes, wpc in ((1, 8), (2, 8), (2, 8)):
    run_sequence(240, 320, lanes, wpc)
