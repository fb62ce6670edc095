"""Multi-GPU sharding of the refinement path.

Windows / directed pairs are independent least-squares problems, so the path shards embarrassingly: one process per
GPU (torch.distributed, backend "nccl" = RCCL on ROCm), a contiguous block of pairs per rank, NO collective on the data
path, and one all_gather of the refined poses (N x 6 floats: bytes-scale, latency only) at the end.  On CPU the same
code runs over "gloo" (tests/test_parallel_cpu.py, world size 2) with a stand-in refine function.

The reference's actual unit of work is a SEQUENCE (run_sequential_optimization.py:186-247: one window after the other over a
KITTI sequence): `refine_sequence_sharded` splits the windows of one sequence contiguously over the ranks -- rank r uploads
only the frames of its block plus the S frames its last window reaches into (the overlap at the seam) -- every rank runs the
library's own window loop (tcsfm_refine_sequence) on its block, and ONE all_gather of [windows, 2S, 6] puts the trajectory
together.  Under the default window rule the result is bit-identical to the single-process sequence whatever the number of
ranks (windows are independent problems and a block is a whole number of calls of `windows_per_call` windows).

Round 5: `refine_dense_sequence_sharded` does the same for the dense mode (BASELINE config 5: pose + per-pixel inverse depth): the poses
are gathered as above, the refined depth maps -- [windows, 2S, 1, H, W] fp32, 31 MB for 64 KITTI windows (SURVEY 8e "optional depth maps")
-- either stay on the rank that refined them (returned with the block's window range: the caller that streams them to disk needs no
collective at all) or are gathered onto every rank with a second all_gather, the caller's choice.
"""
from __future__ import annotations

from typing import Callable, Optional, Tuple

import torch
import torch.distributed as dist


def shard_range(n_items: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous block split: rank r gets [lo, hi); the first n_items % world ranks get one extra item."""
    base, extra = divmod(n_items, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def gather_rows(local: torch.Tensor, n_total: int, group=None) -> torch.Tensor:
    """all_gather of per-rank row blocks [n_r, C] (blocks from shard_range) into the full [n_total, C] on every rank."""
    if not (dist.is_available() and dist.is_initialized()):
        return local
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    nmax = (n_total + world - 1) // world
    pad = torch.zeros((nmax,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    pad[: local.shape[0]] = local
    parts = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(parts, pad, group=group)
    out = []
    for r in range(world):
        lo, hi = shard_range(n_total, r, world)
        out.append(parts[r][: hi - lo])
    return torch.cat(out, 0)


def refine_sharded(refine_fn: Callable[..., torch.Tensor], tensors: dict, n_total: int, group=None) -> torch.Tensor:
    """Every rank refines its block of the batch with `refine_fn(**block)` -> [n_r, 6] and the refined poses of all
    n_total pairs are gathered on every rank.  `tensors` maps argument names to full-batch tensors (first dim n_total);
    a real deployment would only materialise the local block -- the slicing here keeps the helper testable."""
    if dist.is_available() and dist.is_initialized():
        world, rank = dist.get_world_size(group), dist.get_rank(group)
    else:
        world, rank = 1, 0
    lo, hi = shard_range(n_total, rank, world)
    block = {k: v[lo:hi].contiguous() for k, v in tensors.items()}
    local = refine_fn(**block) if hi > lo else torch.zeros((0, 6), dtype=torch.float32, device=next(iter(tensors.values())).device)
    return gather_rows(local, n_total, group)


def sequence_block(n_windows: int, rank: int, world: int, windows_per_call: int = 1) -> Tuple[int, int]:
    """Contiguous block [lo, hi) of a sequence's windows for `rank`, in whole CALLS of `windows_per_call` windows (the library
    refines that many windows per launch sequence; keeping call boundaries where the single-process loop has them makes the
    sharded run bit-identical to it under every window rule)."""
    wpc = max(1, int(windows_per_call))
    lo_c, hi_c = shard_range((n_windows + wpc - 1) // wpc, rank, world)
    return min(lo_c * wpc, n_windows), min(hi_c * wpc, n_windows)


def refine_sequence_sharded(engine, frames: torch.Tensor, depths: torch.Tensor, K, init_poses: torch.Tensor, opts=None, sources: int = 1,
                            windows_per_call: int = 8, target_pos: int = 0, group=None, gather_device: Optional[torch.device] = None,
                            refine_fn: Optional[Callable[..., torch.Tensor]] = None) -> torch.Tensor:
    """The window loop of ONE sequence over all ranks of `group` (run_sequential_optimization.py:186-247 on N GPUs).

    frames [T,3,H,W], depths [T,1,H,W] (CPU, pinned for asynchronous uploads), K [3,3], init_poses [T-S, 2S, 6]; every rank
    passes the same arguments (or at least its own block of them: only rows lo .. hi+S of frames / depths and lo .. hi of
    init_poses are read).  -> refined poses [T-S, 2S, 6] on every rank (CPU tensor).
    gather_device: where the collective runs -- the GPU for RCCL (default when the backend is nccl), the CPU for gloo.
    refine_fn: stand-in for engine.refine_sequence (CPU tests)."""
    T, S = int(frames.shape[0]), int(sources)
    n_win = T - S
    if dist.is_available() and dist.is_initialized():
        world, rank = dist.get_world_size(group), dist.get_rank(group)
    else:
        world, rank = 1, 0
    lo, hi = sequence_block(n_win, rank, world, windows_per_call)
    if hi > lo:
        run = refine_fn or engine.refine_sequence
        local = run(frames[lo:hi + S], depths[lo:hi + S], K, init_poses[lo:hi], opts, sources=S, windows_per_call=windows_per_call,
                    target_pos=target_pos)
        local = torch.as_tensor(local).reshape(hi - lo, 2 * S * 6)
    else:
        local = torch.zeros((0, 2 * S * 6), dtype=torch.float32)
    return _gather_window_blocks(local, n_win, 2 * S * 6, windows_per_call, world, group, gather_device).reshape(n_win, 2 * S, 6)


def _gather_window_blocks(local: torch.Tensor, n_win: int, width: int, windows_per_call: int, world: int, group, gather_device) -> torch.Tensor:
    """all_gather of per-rank blocks [hi - lo, width] (blocks from sequence_block) -> [n_win, width] on every rank (CPU)"""
    if world == 1:
        return local
    if gather_device is None:
        gather_device = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend(group) == "nccl" else torch.device("cpu")
    wpc = max(1, int(windows_per_call))
    n_calls = (n_win + wpc - 1) // wpc
    nmax = ((n_calls + world - 1) // world) * wpc                       # the largest block, in windows
    pad = torch.zeros((nmax, width), dtype=torch.float32, device=gather_device)
    pad[: local.shape[0]] = local.to(gather_device)
    parts = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(parts, pad, group=group)                            # the ONE collective of the job: world x nmax x width floats
    out = []
    for r in range(world):
        rlo, rhi = sequence_block(n_win, r, world, windows_per_call)
        out.append(parts[r][: rhi - rlo].cpu())
    return torch.cat(out, 0)


def odometry_sequence_sharded(net, frames: torch.Tensor, depths: torch.Tensor, K, opts=None, sources: int = 1, iterations: int = 4,
                              windows_per_call: int = 8, target_pos: int = 0, group=None, gather_device: Optional[torch.device] = None,
                              run_fn: Optional[Callable[..., Tuple[torch.Tensor, torch.Tensor]]] = None) -> Tuple[torch.Tensor, torch.Tensor]:
    """tcsfm_odometry_sequence (PoseNet loop + refinement per window, run_sequential_optimization.py:186-247 with
    train_mono.py:64-80 inside) over all ranks of `group`: blocks of whole calls + S overlap frames per rank, as
    refine_sequence_sharded; ONE all_gather carries initial and refined poses together.
    net: posenet.PoseNetHIP of this rank's engine (run_fn: stand-in for net.odometry_sequence in CPU tests).
    -> (initial poses, refined poses), each [T-S, 2S, 6] on every rank (CPU tensors).  The PoseNet's work split depends on the
    images per call only, and a block is a whole number of calls: bit-identical to the single-process run."""
    T, S = int(frames.shape[0]), int(sources)
    n_win = T - S
    if dist.is_available() and dist.is_initialized():
        world, rank = dist.get_world_size(group), dist.get_rank(group)
    else:
        world, rank = 1, 0
    lo, hi = sequence_block(n_win, rank, world, windows_per_call)
    if hi > lo:
        run = run_fn or net.odometry_sequence
        init, out = run(frames[lo:hi + S], depths[lo:hi + S], K, opts, sources=S, iterations=iterations, windows_per_call=windows_per_call,
                        target_pos=target_pos)
        local = torch.cat([torch.as_tensor(init).reshape(hi - lo, 2 * S * 6), torch.as_tensor(out).reshape(hi - lo, 2 * S * 6)], 1)
    else:
        local = torch.zeros((0, 4 * S * 6), dtype=torch.float32)
    both = _gather_window_blocks(local, n_win, 4 * S * 6, windows_per_call, world, group, gather_device)
    return both[:, : 2 * S * 6].reshape(n_win, 2 * S, 6), both[:, 2 * S * 6:].reshape(n_win, 2 * S, 6)


def refine_dense_sequence_sharded(engine, frames: torch.Tensor, depths: torch.Tensor, K, init_poses: torch.Tensor, opts=None, sources: int = 1,
                                  windows_per_call: int = 8, target_pos: int = 0, gather_depths: bool = False, group=None,
                                  gather_device: Optional[torch.device] = None, refine_fn: Optional[Callable[..., Tuple[torch.Tensor, torch.Tensor]]] = None):
    """tcsfm_refine_dense_sequence (the reference's sequential driver with `optimize_depth_pred`: optimizer.py:289-297 returns `depths_opt`,
    run_sequential_optimization.py:195-216 reads `disp_opt`) over all ranks of `group`: blocks of whole calls + S overlap frames per rank as
    refine_sequence_sharded.  Arguments as there.

    -> (poses [T-S, 2S, 6] on every rank, depth maps, (lo, hi)):
       gather_depths = False (default): depth maps = THIS rank's block [hi - lo, 2S, 1, H, W] (CPU) and (lo, hi) says which windows they
         are -- no collective beyond the bytes-scale pose gather (a driver that writes the maps out per rank, as 8 independent writers);
       gather_depths = True: depth maps = all [T-S, 2S, 1, H, W] on every rank -- a second all_gather of world x block x 2S x H x W floats
         (31 MB in all for BASELINE config 3's 64 windows at 640x192; over xGMI a few hundred microseconds), (lo, hi) = (0, T-S).
    Under the default window rule every window is an independent problem and a block is a whole number of calls: poses AND maps are
    bit-identical to the single-process tcsfm_refine_dense_sequence whatever the number of ranks.
    refine_fn: stand-in for engine.refine_dense_sequence (CPU tests).  Unmeasured on multi-GPU hardware (the builder has one GPU)."""
    T, S = int(frames.shape[0]), int(sources)
    n_win = T - S
    if dist.is_available() and dist.is_initialized():
        world, rank = dist.get_world_size(group), dist.get_rank(group)
    else:
        world, rank = 1, 0
    lo, hi = sequence_block(n_win, rank, world, windows_per_call)
    Hh, Ww = int(frames.shape[2]), int(frames.shape[3])
    if hi > lo:
        run = refine_fn or engine.refine_dense_sequence
        local_p, local_d = run(frames[lo:hi + S], depths[lo:hi + S], K, init_poses[lo:hi], opts, sources=S, windows_per_call=windows_per_call,
                               target_pos=target_pos)
        local_p = torch.as_tensor(local_p).reshape(hi - lo, 2 * S * 6)
        local_d = torch.as_tensor(local_d).reshape(hi - lo, 2 * S, 1, Hh, Ww)
    else:
        local_p = torch.zeros((0, 2 * S * 6), dtype=torch.float32)
        local_d = torch.zeros((0, 2 * S, 1, Hh, Ww), dtype=torch.float32)
    poses = _gather_window_blocks(local_p, n_win, 2 * S * 6, windows_per_call, world, group, gather_device).reshape(n_win, 2 * S, 6)
    if not gather_depths or world == 1:
        return poses, local_d, ((lo, hi) if world > 1 else (0, n_win))
    width = 2 * S * Hh * Ww
    maps = _gather_window_blocks(local_d.reshape(hi - lo, width), n_win, width, windows_per_call, world, group, gather_device)
    return poses, maps.reshape(n_win, 2 * S, 1, Hh, Ww), (0, n_win)
