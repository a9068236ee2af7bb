"""N > 1 path on CPU: the strip plan and the gather collective (gloo, world_size 2 and 3).  Each rank packs
the rows it owns out of a known frame exactly as the render kernel packs them (compact mode), the packed shards
are gathered to rank 0 with voxelengine_amd.sharding.gather_frame, and a torch reference of the de-interleave
(the product uses the HIP kernel behind vxrt_deinterleave_strips; that one is covered by the GPU tests) must
rebuild the frame byte for byte."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from voxelengine_amd import sharding


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _reference_deinterleave(plan):
    def fn(shards, frame):
        sh = shards.view(plan.world_size, plan.max_rows, plan.width, 4)
        for r in range(plan.world_size):
            for lr in range(plan.rows_of(r)):
                frame[plan.frame_row(r, lr)] = sh[r, lr]
    return fn


def _worker(rank, world, port, W, H, rows, out_path):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    plan = sharding.ShardPlan(W, H, rows, world, rank)
    g = torch.Generator().manual_seed(1234)
    full = torch.randint(0, 256, (H, W, 4), dtype=torch.uint8, generator=g)   # same on every rank
    local = torch.zeros(plan.shard_bytes, dtype=torch.uint8)
    lv = local.view(plan.max_rows, W, 4)
    for lr in range(plan.local_rows):
        lv[lr] = full[plan.frame_row(rank, lr)]
    shards = torch.zeros((world, plan.shard_bytes), dtype=torch.uint8) if rank == 0 else None
    frame = torch.zeros((H, W, 4), dtype=torch.uint8) if rank == 0 else None
    out = sharding.gather_frame(plan, local, shards, frame, _reference_deinterleave(plan))
    if rank == 0:
        torch.save(dict(ok=bool(torch.equal(out, full))), out_path)
    else:
        assert out is None
    dist.barrier()
    dist.destroy_process_group()


def _pipeline_worker(rank, world, port, W, H, rows, out_path):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    plan = sharding.ShardPlan(W, H, rows, world, rank)
    frame = torch.zeros((H, W, 4), dtype=torch.uint8) if rank == 0 else None
    seen = []

    def deint(shards, fr):
        _reference_deinterleave(plan)(shards, fr)
        seen.append(fr.clone())

    # frame k lands in buffer k mod 2 (what bench.py passes for N > 1, where two render streams alternate); on the CPU the
    # stream context of a step is a no-op
    pair = [frame, torch.zeros_like(frame)] if rank == 0 else [None, None]
    pipe = sharding.GatherPipeline(plan, lambda n: torch.zeros(n, dtype=torch.uint8), pair, deint)
    frames = []
    for k in range(5):
        g = torch.Generator().manual_seed(100 + k)
        full = torch.randint(0, 256, (H, W, 4), dtype=torch.uint8, generator=g)
        frames.append(full)
        with pipe.stream(k):
            buf = pipe.local(k).view(plan.max_rows, W, 4)
            for lr in range(plan.local_rows):
                buf[lr] = full[plan.frame_row(rank, lr)]
            pipe.submit(k)
    pipe.flush()
    if rank == 0:
        ok = len(seen) == 5 and all(torch.equal(a, b) for a, b in zip(seen, frames))
        ok = ok and torch.equal(pipe.frame_of(4), frames[4]) and torch.equal(pipe.frame_of(3), frames[3]) and pipe.frame_of(4) is pair[0]
        torch.save(dict(ok=bool(ok)), out_path)
    dist.barrier()
    dist.destroy_process_group()


def _multi_view_pipeline_worker(rank, world, port, W, H, rows, views, out_path):
    """bench.py's N > 1 step: `views` packed shards back to back per rank, one gather per step."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    plan = sharding.ShardPlan(W, H, rows, world, rank)
    sb = plan.shard_bytes
    frames_out = torch.zeros((views, H, W, 4), dtype=torch.uint8) if rank == 0 else None
    seen = []

    def deint(shards, fr):   # shards: (world, views * shard_bytes); view j of rank r at bytes [j*sb, (j+1)*sb)
        for j in range(views):
            _reference_deinterleave(plan)(shards[:, j * sb:(j + 1) * sb].contiguous(), fr[j])
        seen.append(fr.clone())

    # three buffers deep, as bench.py runs it for N > 1 (7 steps: every buffer is reused at least twice)
    pipe = sharding.GatherPipeline(plan, lambda n: torch.zeros(n, dtype=torch.uint8), frames_out, deint, nbytes=views * sb, depth=3)
    steps = []
    for k in range(7):
        g = torch.Generator().manual_seed(500 + k)
        full = torch.randint(0, 256, (views, H, W, 4), dtype=torch.uint8, generator=g)
        steps.append(full)
        buf = pipe.local(k)
        assert buf.numel() == views * sb
        for j in range(views):
            bj = buf[j * sb:(j + 1) * sb].view(plan.max_rows, W, 4)
            for lr in range(plan.local_rows):
                bj[lr] = full[j, plan.frame_row(rank, lr)]
        pipe.submit(k)
    pipe.flush()
    if rank == 0:
        ok = len(seen) == 7 and all(torch.equal(a, b) for a, b in zip(seen, steps))
        torch.save(dict(ok=bool(ok)), out_path)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,views", [(2, 4), (3, 2)])
def test_gather_pipeline_with_several_views_per_step_gloo(tmp_path, world, views):
    out_path = str(tmp_path / "res.pt")
    mp.spawn(_multi_view_pipeline_worker, args=(world, _free_port(), 32, 40, 8, views, out_path), nprocs=world, join=True)
    assert torch.load(out_path)["ok"]


@pytest.mark.parametrize("world", [2, 3])
def test_two_deep_gather_pipeline_gloo(tmp_path, world):
    """GatherPipeline: frame k's gather overlaps frame k+1's render; buffers are reused only after their
    collective completed; every frame comes out intact and in order."""
    out_path = str(tmp_path / "res.pt")
    mp.spawn(_pipeline_worker, args=(world, _free_port(), 32, 40, 8, out_path), nprocs=world, join=True)
    assert torch.load(out_path)["ok"]


@pytest.mark.parametrize("world,W,H,rows", [(2, 64, 40, 16), (2, 48, 33, 8), (3, 32, 50, 16)])
def test_gather_rebuilds_frame_gloo(tmp_path, world, W, H, rows):
    out_path = str(tmp_path / "res.pt")
    mp.spawn(_worker, args=(world, _free_port(), W, H, rows, out_path), nprocs=world, join=True)
    assert torch.load(out_path)["ok"]


def test_plan_matches_the_c_abi_row_count():
    import voxelengine_amd as vx
    for H, rows, world in [(1080, 16, 8), (1080, 16, 2), (2160, 16, 8), (33, 8, 2), (50, 16, 3), (7, 16, 4)]:
        covered = np.zeros(H, int)
        for r in range(world):
            plan = sharding.ShardPlan(64, H, rows, world, r)
            assert plan.local_rows == vx.compact_rows(H, rows, world, r)
            for lr in range(plan.local_rows):
                covered[plan.frame_row(r, lr)] += 1
        assert (covered == 1).all()          # every frame row belongs to exactly one shard


def test_single_rank_is_a_plain_deinterleave():
    plan = sharding.ShardPlan(16, 20, 8, 1, 0)
    full = torch.arange(20 * 16 * 4, dtype=torch.int64).remainder(251).to(torch.uint8).view(20, 16, 4)
    frame = torch.zeros_like(full)
    sharding.gather_frame(plan, full.reshape(-1).clone(), None, frame, _reference_deinterleave(plan))
    assert torch.equal(frame, full)
