"""frames/s of the C3 clip with every convolution on the fp32 rung (convs="fp32"): Winograd (default) and direct.  python tools/dev/fps_fp32.py"""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import slr_sfs_amd as S
from slr_sfs_amd import pipeline, nets
from bench import smooth_motion, H, W, NFRAMES
dev = torch.device("cuda:0")
torch.manual_seed(0)
image = torch.rand(1, 3, H, W, device=dev) * 2 - 1
motion = torch.from_numpy(smooth_motion(H, W)).to(dev)
m = pipeline.BaselineAnimator(convs="fp32").to(dev).eval()
for wino in ((True,) if "--wino" in sys.argv else ((False,) if "--direct" in sys.argv else (True, False))):
    m.convs = "fp32-winograd" if wino else "fp32"
    m.synthesize(image, motion, NFRAMES)
    torch.cuda.synchronize()
    t = time.perf_counter()
    clip = m.synthesize(image, motion, NFRAMES)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t
    print(f"convs=fp32, winograd={wino}: {NFRAMES / dt:.2f} frames/s ({dt * 1e3 / NFRAMES:.2f} ms per frame)")
