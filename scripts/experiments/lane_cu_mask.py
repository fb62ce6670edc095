"""(needs the measurement hook of appendix R4 in tcsfm_create / tcsfm_set_lanes, not kept in the tree)
ON THE GPU BOX: do lanes whose streams own disjoint slices of the compute units (hipExtStreamCreateWithCUMask) overlap robustly?
One fresh process per case: TCSFM_LANE_CU_PARTITION=0/1 (lanes 1..3 on slices 1..3 of 4; lane 0 = the handle's unrestricted stream),
TCSFM_LANE_CU_INTERLEAVE=0/1 (contiguous / interleaved CU numbering).  Prints the rate of four lanes and of one call at a time."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
CHILD = r'''
import os, sys, time
import numpy as np, torch
sys.path.insert(0, %r)
import os as _os; _os.environ.setdefault("TCSFM_SET_ENV_DEFAULTS", "1")      # (a measurement script owns its process: HIP_FORCE_DEV_KERNARG / GPU_MAX_HW_QUEUES when absent)
from tightly_coupled_sfm_amd import synth
from tightly_coupled_sfm_amd.engine import Engine, default_opts
H, W = 192, 640
t = lambda a: torch.as_tensor(np.ascontiguousarray(a, dtype=np.float32)).cuda()
ws = []
for i in range(12):
    p = synth.make_pair(H, W, seed=300 + i); q = synth.perturb_pose(p["pose_gt"], 300 + i)
    ws.append(dict(tgt=t(p["tgt"][None]), srcs=t(p["src"][None, None]), dt=t(p["depth_t"][None, None]), ds=t(p["depth_s"][None, None, None]), pose=t(np.stack([q, -q])), out=torch.empty(2, 6, device="cuda")))
K = t(synth.make_pair(H, W, seed=300)["K"][None]); torch.cuda.synchronize()
eng = Engine(H, W, 4, lanes=4); o = default_opts(n_iters=4)
def run(nl, n=480):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for k in range(n):
        w = ws[k %% 12]; eng.refine_window_async(k %% nl, w["tgt"], w["srcs"], w["dt"], w["ds"], K, w["pose"], w["out"], o)
    torch.cuda.synchronize(); return n / (time.perf_counter() - t0)
run(4); r4 = max(run(4), run(4)); run(1); r1 = max(run(1), run(1))
eng.set_graph_replay(8); run(4); run(4); g4 = max(run(4), run(4))
print("four lanes %%6.0f (graph replay %%6.0f) | one call at a time %%6.0f" %% (r4, g4, r1))
''' % ROOT
for part, inter in ((0, 0), (1, 0), (1, 1), (0, 0), (1, 0), (1, 1)):
    env = dict(os.environ, TCSFM_LANE_CU_PARTITION=str(part), TCSFM_LANE_CU_INTERLEAVE=str(inter))
    r = subprocess.run([sys.executable, "-c", CHILD], capture_output=True, text=True, env=env, timeout=600)
    print("partition", part, "interleave", inter, "|", (r.stdout.strip().splitlines() or [r.stderr[-300:]])[-1], flush=True)
