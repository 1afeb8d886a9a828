"""Frame sharding + clip assembly on 2 CPU processes (gloo); the GPU path uses the same code
with backend nccl (= RCCL)."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, N, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from slr_sfs_amd import parallel
    mine = parallel.shard_frames(N, rank, world)
    # frame t is filled with the value t (+ channel index / 10)
    local = torch.stack([torch.full((3, 4, 5), float(t)) + torch.arange(3).view(3, 1, 1) / 10 for t in mine]) \
        if mine else torch.zeros(0, 3, 4, 5)
    clip = parallel.gather_clip(local, N, rank, world)
    ok = clip.shape == (N, 3, 4, 5) and all(
        torch.equal(clip[t], torch.full((3, 4, 5), float(t)) + torch.arange(3).view(3, 1, 1) / 10) for t in range(N))
    # round-wise asynchronous assembly (what bench.py uses): same clip
    asm = parallel.ClipAssembler(N, rank, world)
    for f in local:
        asm.push(f)
    clip2 = asm.finish(like=torch.zeros(3, 4, 5))
    ok = ok and clip2.shape == clip.shape and torch.equal(clip2, clip)
    q.put((rank, bool(ok), mine))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("N", [60, 7, 1])
def test_shard_and_gather_world2(N):
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, N, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(ok for _, ok, _ in res)
    covered = sorted(t for _, _, mine in res for t in mine)
    assert covered == list(range(N))                      # every frame rendered exactly once


def test_shard_properties():
    from slr_sfs_amd import parallel
    for N in (60, 61, 5):
        for world in (1, 2, 4, 8):
            shards = [parallel.shard_frames(N, r, world) for r in range(world)]
            assert sorted(sum(shards, [])) == list(range(N))
            assert max(map(len, shards)) == parallel.frames_per_rank(N, world)
            assert max(map(len, shards)) - min(map(len, shards)) <= 1


def test_gather_world1_is_identity():
    from slr_sfs_amd import parallel
    x = torch.randn(5, 3, 2, 2)
    assert parallel.gather_clip(x, 5, 0, 1) is x
    asm = parallel.ClipAssembler(5, 0, 1)
    for f in x:
        asm.push(f)
    assert torch.equal(asm.finish(), x)
