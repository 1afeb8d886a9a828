"""Cost of the round-wise clip assembly on RCCL (one GPU: world_size 1, collective path forced): frames/s of the C3
clip with and without it.  The multi-GPU runs are the driver's; this bounds the per-collective overhead."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))); sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tools")); sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
import os, sys, time, torch, torch.distributed as dist
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import slr_sfs_amd as S
from slr_sfs_amd import parallel
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29555", RANK="0", WORLD_SIZE="1")
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
H, W, N = 768, 1280, 60
torch.manual_seed(0)
an = S.pipeline.BaselineAnimator().cuda().eval()
img = torch.rand(1, 3, H, W, device="cuda") * 2 - 1
m = torch.randn(1, 2, H, W, device="cuda") * 0.5


def plain():
    return an.synthesize(img, m, N)


def assembled():
    asm = parallel.ClipAssembler(N, 0, 1, always_collective=True)
    an.synthesize(img, m, N, on_frame=asm.push)
    return asm.finish()


for name, fn in (("plain", plain), ("round-wise all-gather (RCCL, world 1)", assembled)) * 2:
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f"{name}: {N / dt:.1f} frames/s", flush=True)
dist.destroy_process_group()
