"""Multi-GPU sharding of the refinement path.

Windows / directed pairs are independent least-squares problems, so the path shards embarrassingly: one process per
GPU (torch.distributed, backend "nccl" = RCCL on ROCm), a contiguous block of pairs per rank, NO collective on the data
path, and one all_gather of the refined poses (N x 6 floats: bytes-scale, latency only) at the end.  On CPU the same
code runs over "gloo" (tests/test_parallel_cpu.py, world size 2) with a stand-in refine function.
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
