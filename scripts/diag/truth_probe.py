import sys, numpy as np, torch
sys.path.insert(0, ".")
import os as _os; _os.environ.setdefault("TCSFM_SET_ENV_DEFAULTS", "1")      # (a measurement script owns its process: HIP_FORCE_DEV_KERNARG / GPU_MAX_HW_QUEUES when absent)
from tightly_coupled_sfm_amd import synth, _lib
from tightly_coupled_sfm_amd.engine import Engine, default_opts
H, W = 192, 640
for seed0, noise in ((4242, 0.003), (4242, 0.0), (0, 0.003), (10, 0.003), (20, 0.003)):
    tb = synth.make_batch(2, H, W, seed0=seed0, noise=noise, sampler_consistent=True)
    td = {k: torch.as_tensor(v).cuda().contiguous() for k, v in tb.items()}
    e = Engine(H, W, 2)
    g = td["pose_gt"]
    for name, o in (("gn4", default_opts(n_iters=4)), ("gn8", default_opts(n_iters=8)), ("lm16", default_opts(n_iters=16, solver=_lib.SOLVER_LM))):
        p, _, st = e.refine(td["tgt"], td["src"], td["depth_t"], td["depth_s"], td["K"], td["pose_init"], o, stats=True); e.synchronize()
        et = ((p[:, :3] - g[:, :3]).norm(dim=1) / g[:, :3].norm(dim=1)).cpu().numpy()
        e0 = ((td["pose_init"][:, :3] - g[:, :3]).norm(dim=1) / g[:, :3].norm(dim=1)).cpu().numpy()
        print(seed0, noise, name, "init", np.round(e0, 4), "final", np.round(et, 4), "cost", np.round(st[:, :int(o.n_iters), 0].cpu().numpy(), 5).tolist())
    e.close()
