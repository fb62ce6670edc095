"""N>1 path on CPU: world-size-2 gloo run of the sharding helpers (contiguous split, no data-path collective,
one all_gather of the refined poses).  The GPU refine is replaced by a deterministic stand-in."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, n_total, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from tightly_coupled_sfm_amd import parallel as P
    pose0 = torch.arange(n_total * 6, dtype=torch.float32).reshape(n_total, 6)
    tag = torch.arange(n_total, dtype=torch.float32).reshape(n_total, 1)
    calls = []

    def fake_refine(pose, tag):           # stand-in for Engine.refine: depends only on the pair's own data
        calls.append(pose.shape[0])
        return pose * 2 + tag

    out = P.refine_sharded(fake_refine, dict(pose=pose0, tag=tag), n_total)
    lo, hi = P.shard_range(n_total, rank, world)
    ok = torch.equal(out, pose0 * 2 + tag) and calls == ([hi - lo] if hi > lo else [])
    ret[rank] = (bool(ok), lo, hi)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("n_total", [8, 5, 1])
def test_sharded_refine_gloo_world2(n_total):
    world, port = 2, _free_port()
    with mp.Manager() as m:
        ret = m.dict()
        mp.spawn(_worker, args=(world, port, n_total, ret), nprocs=world, join=True)
        ranges = [ret[r][1:] for r in range(world)]
        assert all(ret[r][0] for r in range(world))
        assert ranges[0][0] == 0 and ranges[-1][1] == n_total and ranges[0][1] == ranges[1][0]   # disjoint, covering


def test_shard_range_properties():
    from tightly_coupled_sfm_amd.parallel import shard_range
    for n in (0, 1, 7, 64, 65):
        for w in (1, 2, 3, 8):
            blocks = [shard_range(n, r, w) for r in range(w)]
            assert blocks[0][0] == 0 and blocks[-1][1] == n
            assert all(blocks[i][1] == blocks[i + 1][0] for i in range(w - 1))
            sizes = [hi - lo for lo, hi in blocks]
            assert max(sizes) - min(sizes) <= 1


def _seq_worker(rank, world, port, T, S, wpc, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from tightly_coupled_sfm_amd import parallel as P
    g = torch.Generator().manual_seed(0)
    frames = torch.rand((T, 3, 4, 6), generator=g); depths = torch.rand((T, 1, 4, 6), generator=g) + 0.5
    init = torch.rand((T - S, 2 * S, 6), generator=g)
    seen = []

    def fake_sequence(fr, dp, K, p0, opts, sources, windows_per_call, target_pos):
        # a window's result depends on exactly its own S + 1 frames and its initial poses -- like the real loop
        seen.append((fr.shape[0], p0.shape[0]))
        w = torch.stack([fr[i:i + sources + 1].sum() + dp[i:i + sources + 1].sum() for i in range(p0.shape[0])])
        return p0 * 2 + w[:, None, None]

    out = P.refine_sequence_sharded(None, frames, depths, None, init, None, sources=S, windows_per_call=wpc, refine_fn=fake_sequence)
    full = fake_sequence(frames, depths, None, init, None, S, wpc, 0)
    lo, hi = P.sequence_block(T - S, rank, world, wpc)
    ok = torch.equal(out, full) and (seen[0] == (hi - lo + S, hi - lo) if hi > lo else len(seen) == 1)

    def fake_odometry(fr, dp, K, opts, sources, iterations, windows_per_call, target_pos):       # (initial, refined) per window
        n = fr.shape[0] - sources
        w = torch.stack([fr[i:i + sources + 1].sum() * iterations + dp[i:i + sources + 1].sum() for i in range(n)])
        i0 = w[:, None, None] * torch.ones((n, 2 * sources, 6))
        return i0, i0 * 3 + 1

    i_sh, o_sh = P.odometry_sequence_sharded(None, frames, depths, None, None, sources=S, iterations=3, windows_per_call=wpc, run_fn=fake_odometry)
    i_1, o_1 = fake_odometry(frames, depths, None, None, S, 3, wpc, 0)
    ok = ok and torch.equal(i_sh, i_1) and torch.equal(o_sh, o_1)
    ret[rank] = (bool(ok), lo, hi)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("T,S,wpc", [(21, 1, 8), (12, 2, 4), (4, 2, 8), (30, 1, 1)])
def test_sequence_sharded_gloo_world2(T, S, wpc):
    """the window loop of ONE sequence split over two ranks: contiguous blocks in whole calls, S overlap frames at the seam, one
    all_gather -- equal to the single-process loop"""
    world, port = 2, _free_port()
    with mp.Manager() as m:
        ret = m.dict()
        mp.spawn(_seq_worker, args=(world, port, T, S, wpc, ret), nprocs=world, join=True)
        assert all(ret[r][0] for r in range(world)), dict(ret)
        assert ret[0][1] == 0 and ret[0][2] == ret[1][1] and ret[1][2] == T - S and (ret[0][2] % wpc == 0 or ret[0][2] == T - S)


def _dense_seq_worker(rank, world, port, T, S, wpc, gather, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from tightly_coupled_sfm_amd import parallel as P
    g = torch.Generator().manual_seed(1)
    Hh, Ww = 4, 6
    frames = torch.rand((T, 3, Hh, Ww), generator=g); depths = torch.rand((T, 1, Hh, Ww), generator=g) + 0.5
    init = torch.rand((T - S, 2 * S, 6), generator=g)

    def fake_dense(fr, dp, K, p0, opts, sources, windows_per_call, target_pos):
        # poses and maps of a window depend on exactly its own S + 1 frames and initial poses -- like the real loop
        n = p0.shape[0]
        w = torch.stack([fr[i:i + sources + 1].sum() + dp[i:i + sources + 1].sum() for i in range(n)])
        maps = torch.stack([torch.stack([dp[i + (j % (sources + 1))] * (1.0 + 0.01 * j) + w[i] for j in range(2 * sources)]) for i in range(n)])
        return p0 * 2 + w[:, None, None], maps

    poses, maps, (lo_r, hi_r) = P.refine_dense_sequence_sharded(None, frames, depths, None, init, None, sources=S, windows_per_call=wpc,
                                                                gather_depths=gather, refine_fn=fake_dense)
    full_p, full_d = fake_dense(frames, depths, None, init, None, S, wpc, 0)
    lo, hi = P.sequence_block(T - S, rank, world, wpc)
    ok = torch.equal(poses, full_p)
    if gather:
        ok = ok and (lo_r, hi_r) == (0, T - S) and torch.equal(maps, full_d)
    else:
        ok = ok and (lo_r, hi_r) == (lo, hi) and torch.equal(maps, full_d[lo:hi]) and tuple(maps.shape) == (hi - lo, 2 * S, 1, Hh, Ww)
    ret[rank] = (bool(ok), lo, hi)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("T,S,wpc,gather", [(21, 1, 8, False), (21, 1, 8, True), (12, 2, 4, True), (4, 2, 8, False), (30, 1, 1, True)])
def test_dense_sequence_sharded_gloo_world2(T, S, wpc, gather):
    """VERDICT r04 #6 / SURVEY 8e: the dense mode of a sequence over two ranks -- poses by all_gather, depth maps left per rank with their
    window range or gathered (the caller's choice) -- equal to the single-process loop bit for bit"""
    world, port = 2, _free_port()
    with mp.Manager() as m:
        ret = m.dict()
        mp.spawn(_dense_seq_worker, args=(world, port, T, S, wpc, gather, ret), nprocs=world, join=True)
        assert all(ret[r][0] for r in range(world)), dict(ret)
        assert ret[0][1] == 0 and ret[0][2] == ret[1][1] and ret[1][2] == T - S
