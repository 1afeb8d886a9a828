"""Child process of tests/test_gpu_parity.py::test_two_ranks_one_clip: rank RANK of WORLD_SIZE renders its share of one
clip on cuda:0 (both ranks share the one GPU of the test box; the transport is gloo on device tensors) exactly as
bench.py does on N GPUs -- encoder in row bands, frames round-robin, round-wise asynchronous assembly -- and compares
the assembled clip with the single-process clip."""
import os
import sys

import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import slr_sfs_amd as S  # noqa: E402
from slr_sfs_amd import parallel  # noqa: E402

rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
torch.cuda.set_device(0)
dist.init_process_group("gloo", rank=rank, world_size=world)
H, W, N = 48, 72, 7
for name, cls in (("baseline", S.pipeline.BaselineAnimator), ("slr-v1", S.pipeline.SLRv1Animator)):
    torch.manual_seed(4)                                       # same weights, image and motion on every rank
    an = cls().cuda().eval()
    img = torch.rand(1, 3, H, W, device="cuda") * 2 - 1
    m = torch.randn(1, 2, H, W, device="cuda") * 1.5
    ref = an.synthesize(img, m, N)
    mine = parallel.shard_frames(N, rank, world)
    asm = parallel.ClipAssembler(N, rank, world)
    an.synthesize(img, m, N, frames=mine, on_frame=asm.push, shard=(rank, world))
    clip = asm.finish(like=img[0])
    torch.cuda.synchronize()
    err = (clip - ref).abs().max().item()
    assert clip.shape == ref.shape and err < 1e-4, (name, rank, err)
    one = parallel.gather_clip(an.synthesize(img, m, N, frames=mine), N, rank, world)      # the one-collective form
    assert (one - ref).abs().max().item() < 1e-4, (name, rank)
dist.barrier()
dist.destroy_process_group()
print(f"RANK{rank} OK")
