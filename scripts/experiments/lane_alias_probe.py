"""ON THE GPU BOX: is the destructive overlap of lanes an ADDRESS-aliasing effect?  In a process layout where four lanes are in the bad state
(examples/queued_windows.py's), the lanes are created one by one with a dummy device allocation of a given size in between (so that every
lane's scratch sits at a different offset); sizes: none, powers of two, odd multiples of 4 KB.  One fresh process per case."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
CHILD = r'''
import os, sys, time
import numpy as np, torch
sys.path.insert(0, %r)
import os as _os; _os.environ.setdefault("TCSFM_SET_ENV_DEFAULTS", "1")      # (a measurement script owns its process: HIP_FORCE_DEV_KERNARG / GPU_MAX_HW_QUEUES when absent)
from tightly_coupled_sfm_amd import synth
from tightly_coupled_sfm_amd.engine import Engine, default_opts
pad = int(os.environ["PAD"])
H, W = 192, 640
t = lambda a: torch.as_tensor(np.ascontiguousarray(a, dtype=np.float32)).cuda()
ws = []
for i in range(12):
    p = synth.make_pair(H, W, seed=300 + i); q = synth.perturb_pose(p["pose_gt"], 300 + i)
    ws.append(dict(tgt=t(p["tgt"][None]), srcs=t(p["src"][None, None]), dt=t(p["depth_t"][None, None]), ds=t(p["depth_s"][None, None, None]), pose=t(np.stack([q, -q])), out=torch.empty(2, 6, device="cuda")))
K = t(synth.make_pair(H, W, seed=300)["K"][None]); torch.cuda.synchronize()
eng = Engine(H, W, 4, lanes=1); keep = []
for l in (2, 3, 4):
    if pad: keep.append(torch.empty(pad * l, dtype=torch.uint8, device="cuda"))
    eng.set_lanes(l)
o = default_opts(n_iters=4)
def run(nl, n=480):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for k in range(n):
        w = ws[k %% 12]; eng.refine_window_async(k %% nl, w["tgt"], w["srcs"], w["dt"], w["ds"], K, w["pose"], w["out"], o)
    torch.cuda.synchronize(); return n / (time.perf_counter() - t0)
run(4); r4 = max(run(4), run(4)); run(1); r1 = max(run(1), run(1))
print("four lanes %%6.0f | one call at a time %%6.0f" %% (r4, r1))
''' % ROOT
for pad in (0, 1 << 20, 1 << 24, 4096 * 37, 4096 * 1021, 4096 * 2579 + 512, 3 * (1 << 20) + 4096 * 7):
    r = subprocess.run([sys.executable, "-c", CHILD], capture_output=True, text=True, env=dict(os.environ, PAD=str(pad)), timeout=600)
    print("pad bytes x lane", pad, "|", (r.stdout.strip().splitlines() or [r.stderr[-300:]])[-1], flush=True)
