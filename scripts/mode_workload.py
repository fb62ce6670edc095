"""Short workloads of the NON-headline modes for rocprofv3 passes (kernel stats / PMC): python scripts/mode_workload.py <mode> [calls]
   dense320 | dense448 : BASELINE config 5 (fwd+inv pair, 4 GN its) at 320x240 / 448x256      -> k_dense_linearize
   dense_sat           : the same kernel with the chip full (64 directed pairs at 320x240)
   scale7              : BASELINE config 4 (pose + depth scale, 8 its, 640x192)                 -> k_linearize<7>, k_solve<7>
   window_sel          : the reference's KITTI default window B=1, S=2, min over sources, w_dc  -> k_linearize<6,true,..,SEL>
   shard8              : BASELINE config 3's per-GPU shard, 8 windows = 16 directed pairs
   window_ref          : the KITTI window under the REFERENCE window rule                       -> k_linearize<..SEL> + k_solve group normalisers
   joint_kitti / copies_kitti : the KITTI window in the dense mode, joint (shared depth, 12x12) / per-pair copies"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import os as _os; _os.environ.setdefault("TCSFM_SET_ENV_DEFAULTS", "1")      # (a measurement script owns its process: HIP_FORCE_DEV_KERNARG / GPU_MAX_HW_QUEUES when absent)
from tightly_coupled_sfm_amd import synth
from tightly_coupled_sfm_amd.engine import Engine, default_opts

mode = sys.argv[1]
calls = int(sys.argv[2]) if len(sys.argv) > 2 else 20


def dev_batch(n, H, W):
    b = synth.make_batch(n, H, W, seed0=0, both_directions=True)
    return {k: torch.as_tensor(v).cuda().contiguous() for k, v in b.items()}


if mode in ("dense320", "dense448", "dense_sat"):
    H, W = (256, 448) if mode == "dense448" else (240, 320)
    n = 64 if mode == "dense_sat" else 2
    d = dev_batch(2, H, W)
    if n > 2:
        d = {k: v.repeat((n // 2,) + (1,) * (v.dim() - 1)).contiguous() for k, v in d.items()}
    e = Engine(H, W, n)
    o = default_opts(n_iters=4, min_depth=0.03, max_depth=3.0)
    for _ in range(calls):
        out = e.refine_dense(d["tgt"], d["src"], d["depth_t"], d["depth_s"], d["K"], d["pose_init"], o)
elif mode == "scale7":
    H, W = 192, 640
    d = dev_batch(2, H, W)
    e = Engine(H, W, 2)
    for _ in range(calls):
        out = e.refine(d["tgt"], d["src"], d["depth_t"], d["depth_s"], d["K"], d["pose_init"], default_opts(n_iters=8, refine=1))
elif mode == "window_sel":
    H, W, B, S = 192, 640, 1, 2
    b = synth.make_batch(2 * S, H, W, seed0=0)
    d = {k: torch.as_tensor(v).cuda().contiguous() for k, v in b.items()}
    tgt, srcs = d["tgt"][:1], d["src"][:S].reshape(S, B, 3, H, W)
    dt, ds = d["depth_t"][:1], d["depth_s"][:S].reshape(S, B, 1, H, W)
    pose = torch.cat([d["pose_init"][:S], -d["pose_init"][:S]]).contiguous()
    e = Engine(H, W, 2 * S * B)
    for _ in range(calls):
        out = e.refine_window(tgt, srcs, dt, ds, d["K"][:1], pose, default_opts(n_iters=4, w_dc=0.15), argmin=True)
elif mode in ("window_ref", "joint_kitti", "copies_kitti"):
    # the KITTI window (B=1, S=2, min over sources): pose mode under the REFERENCE window rule / dense mode joint / dense per-pair copies
    H, W, B, S = 192, 640, 1, 2
    b = synth.make_batch(2 * S, H, W, seed0=0)
    d = {k: torch.as_tensor(v).cuda().contiguous() for k, v in b.items()}
    tgt, srcs = d["tgt"][:1], d["src"][:S].reshape(S, B, 3, H, W)
    dt, ds = d["depth_t"][:1], d["depth_s"][:S].reshape(S, B, 1, H, W)
    pose = torch.cat([d["pose_init"][:S], -d["pose_init"][:S]]).contiguous()
    e = Engine(H, W, 2 * S * B)
    for _ in range(calls):
        if mode == "window_ref":
            out = e.refine_window(tgt, srcs, dt, ds, d["K"][:1], pose, default_opts(n_iters=4, w_dc=0.15, window_rule=1), argmin=True)
        else:
            out = e.refine_dense_window(tgt, srcs, dt, ds, d["K"][:1], pose, default_opts(n_iters=4, dense_joint=1 if mode == "joint_kitti" else 0), argmin=True)
elif mode == "shard8":
    H, W = 192, 640
    d = dev_batch(16, H, W)
    e = Engine(H, W, 16)
    for _ in range(calls):
        out = e.refine(d["tgt"], d["src"], d["depth_t"], d["depth_s"], d["K"], d["pose_init"], default_opts(n_iters=4))
else:
    raise SystemExit(f"unknown mode {mode}")
torch.cuda.synchronize()
print("done", mode, float(out[0].abs().sum()))
