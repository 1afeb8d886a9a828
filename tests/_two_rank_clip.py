"""Child process of tests/test_gpu_parity.py::test_two_ranks_one_clip / test_two_ranks_on_rccl: rank RANK of WORLD_SIZE
renders its share of one clip exactly as bench.py / runner.py do on N GPUs -- frames round-robin, both assembly forms (ONE
all-gather of the finished clip = the north_star form; round-wise asynchronous all-gathers), encoder redundant or in row
bands -- and compares the assembled clip with the single-process clip.
SLR_TEST_BACKEND=gloo (default): every rank on cuda:0 (the test box has one GPU; gloo moves the device tensors);
SLR_TEST_BACKEND=nccl: rank r on cuda:r, collectives over RCCL (needs >= WORLD_SIZE GPUs)."""
import os
import sys

import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import slr_sfs_amd as S  # noqa: E402
from slr_sfs_amd import parallel  # noqa: E402

rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
backend = os.environ.get("SLR_TEST_BACKEND", "gloo")
if backend == "nccl":
    torch.cuda.set_device(rank)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", rank))
else:
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
H, W = 48, 72
for name, cls in (("baseline", S.pipeline.BaselineAnimator), ("slr-v1", S.pipeline.SLRv1Animator)):
    for N in (7, 1):                                           # N = 1 < world: rank 1 renders nothing and must still enter
        torch.manual_seed(4)                                   # every collective; same weights, image, motion on every rank
        an = cls().cuda().eval()
        img = torch.rand(1, 3, H, W, device="cuda") * 2 - 1
        m = torch.randn(1, 2, H, W, device="cuda") * 1.5
        ref = an.synthesize(img, m, N)
        mine = parallel.shard_frames(N, rank, world)
        # round-wise asynchronous assembly + encoder in row bands
        asm = parallel.ClipAssembler(N, rank, world)
        an.synthesize(img, m, N, frames=mine, on_frame=asm.push, shard=(rank, world))
        clip = asm.finish(like=img[0])
        torch.cuda.synchronize()
        err = (clip - ref).abs().max().item()
        assert clip.shape == ref.shape and err < 1e-4, (name, N, rank, err)
        # the north_star form: redundant encoder, ONE all-gather of the finished clip (uneven shards: N = 7 over 2 ranks)
        one = parallel.gather_clip(an.synthesize(img, m, N, frames=mine), N, rank, world)
        assert one.shape == ref.shape and (one - ref).abs().max().item() < 1e-4, (name, N, rank)
        # ... with the frames converted to uint8 by every rank before the collective (what the writer saves: 1 byte per sample moved)
        from slr_sfs_amd import io
        u8 = parallel.gather_clip(parallel.frames_for_assembly(an.synthesize(img, m, N, frames=mine)), N, rank, world)
        want = io.frames_to_uint8(ref)
        assert u8.dtype == torch.uint8 and u8.shape == want.shape and int((u8.int() - want.int()).abs().max()) <= 1, (name, N, rank)
        if name == "slr-v1":                                   # the runner's dict of outputs, every key gathered by every rank
            outs = an.synthesize(img, m, N, frames=mine, shard=(rank, world), keys=cls.KEYS)
            full = an.synthesize(img, m, N, keys=cls.KEYS)
            for k in cls.KEYS:
                got = outs[k] if k == "BGImg" else parallel.gather_clip(outs[k].contiguous(), N, rank, world)
                assert got.shape == full[k].shape and (got - full[k]).abs().max().item() < 1e-4, (k, N, rank)
        # the banded encoder by itself on device buffers
        fs = parallel.encode_banded(an.encoder, img, rank, world)
        fs0 = an.encoder(img)
        for got, want in zip(fs if isinstance(fs, tuple) else (fs,), fs0 if isinstance(fs0, tuple) else (fs0,)):
            assert torch.equal(got, want), (name, rank)
rep = parallel.communicator_report(torch.device("cuda", torch.cuda.current_device()), 7, 1.0)
assert rep["world_size"] == world and rep["backend"] == backend and [r["rank"] for r in rep["ranks"]] == list(range(world))
if backend == "nccl":
    assert rep["distinct_devices"] == world and rep["rccl_version"], rep
dist.barrier()
dist.destroy_process_group()
print(f"RANK{rank} OK")
