"""Frame sharding + clip assembly on 2, 4 and 8 CPU processes (gloo); the GPU path uses the same code
with backend nccl (= RCCL).  World 8 / 4 with N = 60 is config C5's shape: 60 = 7 * 8 + 4, so ranks 4-7 have no frame in
the last round (their padding must land behind frame 59 and be cut off)."""
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


@pytest.mark.parametrize("world,N", [(2, 60), (2, 7), (2, 1), (4, 60), (8, 60), (8, 5)])
def test_shard_and_gather(world, N):
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, N, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=300) for _ in procs]
    for p in procs:
        p.join(timeout=120)
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


def _enc_worker(rank, world, port, q, hw=(45, 40)):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(1)
    from slr_sfs_amd import nets, parallel
    torch.manual_seed(5)                                  # same weights and image on every rank
    ok = True
    with nets.cpu_reference(), torch.no_grad():
        img = torch.rand(1, 3, *hw) * 2 - 1               # 45 rows: bands of 23 + 22, padded for the gather
        for enc in (nets.EncoderWithZ().eval(), nets.Encoder(3, 2).eval()):
            want = enc(img)
            got = parallel.encode_banded(enc, img, rank, world)
            want = want if isinstance(want, tuple) else (want,)
            got = got if isinstance(got, tuple) else (got,)
            ok = ok and len(want) == len(got) and all(
                g.shape == w.shape and g.is_contiguous() and torch.equal(g, w) for g, w in zip(got, want))
    q.put((rank, bool(ok), []))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,hw", [(2, (45, 40)), (4, (768, 8)), (8, (768, 8))])
def test_banded_encoder(world, hw):
    """The frame-invariant encoder in row bands + all-gather == the encoder on the whole image (CPU definition of the
    networks; the device kernels: tests/test_gpu_parity.py::test_banded_encoder_is_exact).  768 rows over 8 / 4 ranks:
    bands of 96 / 192 rows with the 16-row halo, the working height of config C5."""
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_enc_worker, args=(r, world, port, q, hw)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=400) for _ in procs]
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert all(ok for _, ok, _ in res)


def test_banded_encoder_refuses_empty_bands_on_every_rank():
    """More ranks than bands of ceil(H/world) rows: EVERY rank raises before any work or collective (a rank that
    asserted alone would leave the others waiting in the all-gather)."""
    from slr_sfs_amd import nets, parallel
    img = torch.zeros(1, 3, 72, 8)
    enc = nets.Encoder(3, 2)
    for rank in (0, 7, 15):
        with pytest.raises(ValueError, match="non-empty bands"):
            parallel.encode_banded(enc, img, rank, 16)


def test_band_rows_cover_the_image():
    from slr_sfs_amd import parallel
    for H in (768, 45, 8):
        for world in (1, 2, 3, 8):
            bands = [parallel.band_rows(H, r, world) for r in range(world)]
            assert bands[0][0] == 0 and bands[-1][1] == H
            assert all(a[1] == b[0] for a, b in zip(bands, bands[1:]))
