#!/usr/bin/env python3
"""The reference's sequential driver (optimization_experiments/run_sequential_optimization.py:186-247: one window after the
other over a sequence) on N GPUs: the windows of ONE sequence are split contiguously over the ranks (S overlap frames at the
seams), every rank runs the library's window loop on its block, one all_gather of [windows, 2S, 6] returns the trajectory.

    python examples/run_sequence_sharded.py --frames 64                                     (1 GPU)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 \\
        examples/run_sequence_sharded.py --frames 512                                       (8 GPUs, RCCL over xGMI)

Synthetic frames (tightly_coupled_sfm_amd.synth); --dump writes the gathered poses (rank 0) for comparison with a
single-process run.  TCSFM_BENCH_BACKEND=gloo / TCSFM_BENCH_ONE_DEVICE=1: rehearsal of several ranks on one card (tests).
"""
import argparse
import os
import sys
import time

import numpy as np
import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import os as _os; _os.environ.setdefault("TCSFM_SET_ENV_DEFAULTS", "1")      # (a measurement script owns its process: HIP_FORCE_DEV_KERNARG / GPU_MAX_HW_QUEUES when absent)
from tightly_coupled_sfm_amd import parallel, synth                               # noqa: E402
from tightly_coupled_sfm_amd.engine import Engine, default_opts                   # noqa: E402


def make_sequence(T, H, W, S, seed=0):
    """T frames = T independent synthetic (target, source) scenes chained: frame k is scene k's target; window w's initial poses
    are small perturbations of zero motion -- enough for a deterministic workload whose windows differ"""
    rng = np.random.default_rng(seed)
    b = synth.make_batch(T, H, W, seed0=seed)
    frames = torch.as_tensor(b["tgt"]).contiguous().pin_memory()
    depths = torch.as_tensor(b["depth_t"]).contiguous().pin_memory()
    init = torch.as_tensor(rng.normal(scale=[1e-3, 1e-3, 2e-3, 3e-4, 3e-4, 3e-4], size=(T - S, 2 * S, 6)).astype(np.float32))
    return frames, depths, b["K"][0], init


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=64)
    ap.add_argument("--sources", type=int, default=1)
    ap.add_argument("--height", type=int, default=192)
    ap.add_argument("--width", type=int, default=640)
    ap.add_argument("--windows-per-call", type=int, default=8)
    ap.add_argument("--gn-iters", type=int, default=4)
    ap.add_argument("--lanes", type=int, default=2)
    ap.add_argument("--dump", default="")
    ap.add_argument("--dense", action="store_true", help="pose + per-pixel inverse depth (tcsfm_refine_dense_sequence per rank): poses gathered, depth "
                    "maps left on the rank that refined them (--gather-depths: gathered onto every rank as well)")
    ap.add_argument("--gather-depths", action="store_true")
    ap.add_argument("--dump-depths", default="", help="--dense: rank r writes its block of depth maps to <this>.<lo>-<hi>.npy (gathered: rank 0 writes all)")
    ap.add_argument("--odometry", type=int, default=0, metavar="ITERS",
                    help="initial poses from the coupled PoseNet loop (ITERS network evaluations per window, seeded stand-in weights) instead of --init")
    args = ap.parse_args()
    world, rank, local_rank = int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0"))
    backend = os.environ.get("TCSFM_BENCH_BACKEND", "nccl")
    if os.environ.get("TCSFM_BENCH_ONE_DEVICE"):
        local_rank = 0
    torch.cuda.set_device(local_rank)
    if world > 1 or os.environ.get("TCSFM_FORCE_DIST"):
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29517")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    S, H, W = args.sources, args.height, args.width
    frames, depths, K, init = make_sequence(args.frames, H, W, S)
    eng = Engine(H, W, 2 * S * args.windows_per_call, lanes=args.lanes)
    o = default_opts(n_iters=args.gn_iters)
    if args.odometry:
        sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
        import standins                                                           # seeded PoseNet parameters (a real run loads a checkpoint)
        from tightly_coupled_sfm_amd.posenet import PoseNetHIP
        net = PoseNetHIP(eng, 2 * S * args.windows_per_call, standins.posenet_params(0))
        run = lambda: parallel.odometry_sequence_sharded(net, frames, depths, K, o, sources=S, iterations=args.odometry,
                                                         windows_per_call=args.windows_per_call)[1]
    elif args.dense:
        o = default_opts(n_iters=args.gn_iters, min_depth=0.03, max_depth=3.0)
        last = {}
        def run():
            poses, maps, blk = parallel.refine_dense_sequence_sharded(eng, frames, depths, K, init, o, sources=S, windows_per_call=args.windows_per_call,
                                                                      gather_depths=args.gather_depths)
            last["maps"], last["blk"] = maps, blk
            return poses
    else:
        run = lambda: parallel.refine_sequence_sharded(eng, frames, depths, K, init, o, sources=S, windows_per_call=args.windows_per_call)
    run()      # warm-up
    torch.cuda.synchronize()
    if dist.is_initialized():
        dist.barrier()
    t0 = time.perf_counter()
    poses = run()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if args.dense and args.dump_depths and (rank == 0 or not args.gather_depths):
        lo, hi = last["blk"]
        np.save(f"{args.dump_depths}.{lo}-{hi}.npy", last["maps"].numpy())
    if rank == 0:
        assert torch.isfinite(poses).all() and tuple(poses.shape) == (args.frames - S, 2 * S, 6)
        if args.dump:
            np.save(args.dump, poses.numpy())
        print(f"{args.frames - S} windows of {W}x{H} (S={S}) on {world} rank(s): {dt * 1e3:.1f} ms = {(args.frames - S) / dt:.0f} windows/s "
              f"(PCIe-inclusive, gather included)", flush=True)
    if dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
