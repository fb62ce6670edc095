"""What a per-PAIR linearisation kernel costs against k_dense_joint at the chip-filling shape of the reference driver's minibatch (6 KITTI windows = 12 forward
pair evaluations at 192x640): k_linearize<7> (pose + one extra column: the pose + depth-scale mode's kernel, whose seventh column is structurally the inverse-depth
column of the dense mode, reduced instead of written per pixel) on 12 / 64 pairs per launch, in-kernel bracket, beside k_dense_joint<2, REF>'s bracket for 6 targets.
A design note for the next round, not a product path."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
os.environ.setdefault("TCSFM_SET_ENV_DEFAULTS", "1")
import torch
from tightly_coupled_sfm_amd import synth, _lib
from tightly_coupled_sfm_amd.engine import Engine, default_opts
import test_gpu_dense_reference as T
H, W = 192, 640


def lin_us(e, step, n=30):
    for _ in range(10): step()
    torch.cuda.synchronize()
    e.profile_begin()
    for _ in range(n): step()
    pr = e.profile_end()
    return pr["linearize_kernel"][0] / max(pr["linearize_kernel"][1], 1) * 1e3


b = synth.make_batch(2, H, W, seed0=0, both_directions=True)
dev = {k: torch.as_tensor(v).cuda().contiguous() for k, v in b.items()}
for rep, tag in ((6, "12 pairs"), (32, "64 pairs")):
    big = {k: dev[k].repeat((rep,) + (1,) * (dev[k].dim() - 1)).contiguous() for k in ("tgt", "src", "depth_t", "depth_s", "K", "pose_init")}
    e = Engine(H, W, 2 * rep)
    out = torch.empty_like(big["pose_init"])
    for name, o in (("k_linearize<6> (pose)", default_opts(n_iters=4)), ("k_linearize<7> (pose + one more column)", default_opts(n_iters=4, refine=_lib.REFINE_POSE_SCALE)),
                    ("k_linearize<6, DC> (pose + depth-consistency term)", default_opts(n_iters=4, w_dc=0.15))):
        us = lin_us(e, lambda: e.refine_into(big["tgt"], big["src"], big["depth_t"], big["depth_s"], big["K"], big["pose_init"], out, o))
        print(json.dumps({"kernel": name, "pairs_per_launch": 2 * rep, "launch_us": round(us, 2), "us_per_pair": round(us / (2 * rep), 3)}), flush=True)
    # the pair-form DENSE kernel (adjoint form of the SSIM gradient, per-pixel inverse depth eliminated per pair; 2-pixel halo): what a per-pair forward kernel
    # of the reference-loss mode can realistically be -- the per-pixel depth column needs the adjoint gather, which k_linearize's pass B does not have
    od = default_opts(n_iters=4, min_depth=0.06, max_depth=2.67)
    us = lin_us(e, lambda: e.refine_dense(big["tgt"], big["src"], big["depth_t"], big["depth_s"], big["K"], big["pose_init"], od))
    print(json.dumps({"kernel": "k_dense_linearize (pair-form dense, adjoint form)", "pairs_per_launch": 2 * rep, "launch_us": round(us, 2), "us_per_pair": round(us / (2 * rep), 3)}), flush=True)
    e.close()
B, S = 6, 2
w = T._window(B, S, H, W, seed=31)
t = {k: T._dev(v) for k, v in w.items()}
dt4, ds5 = t["depth_t"][:, None].contiguous(), t["depth_s"][:, :, None].contiguous()
e = Engine(H, W, 2 * S * B)
o = default_opts(n_iters=4, w_dc=0.15, prior_init=0.1, min_depth=0.06, max_depth=2.67, window_rule=_lib.WINDOW_REFERENCE)
us = lin_us(e, lambda: e.refine_dense_window(t["tgt"], t["srcs"], dt4, ds5, t["K"], t["pose"], o, argmin=True))
print(json.dumps({"kernel": "k_dense_joint<2, REF> (6 targets x 2 sources = 12 forward pair evaluations)", "pairs_per_launch": 12, "launch_us": round(us, 2), "us_per_pair": round(us / 12, 3)}), flush=True)
