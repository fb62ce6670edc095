"""Chip-filling workload for PMC passes: 32 windows (64 directed pairs) per refine call, a few calls."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import os as _os; _os.environ.setdefault("TCSFM_SET_ENV_DEFAULTS", "1")      # (a measurement script owns its process: HIP_FORCE_DEV_KERNARG / GPU_MAX_HW_QUEUES when absent)
from tightly_coupled_sfm_amd import synth
from tightly_coupled_sfm_amd.engine import Engine, default_opts
H, W, REP = 192, 640, int(sys.argv[1]) if len(sys.argv) > 1 else 32
b = synth.make_batch(2, H, W, seed0=0, both_directions=True)
dev = {k: torch.as_tensor(v).cuda().contiguous() for k, v in b.items()}
big = {k: dev[k].repeat((REP,) + (1,) * (dev[k].dim() - 1)).contiguous() for k in ("tgt", "src", "depth_t", "depth_s", "K", "pose_init")}
e = Engine(H, W, 2 * REP)
out = torch.empty_like(big["pose_init"])
o = default_opts(n_iters=4)
for _ in range(40):        # enough launches for steady clocks: rocprofv3's average of a 6-call run read 25 % high
    e.refine_into(big["tgt"], big["src"], big["depth_t"], big["depth_s"], big["K"], big["pose_init"], out, o)
torch.cuda.synchronize()
print("done", float(out.abs().sum()))
