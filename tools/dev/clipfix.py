import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))); sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tools")); sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
import os, sys, time
import numpy as np, torch
pass
import bench
from slr_sfs_amd import pipeline
H, W = 768, 1280
dev = torch.device("cuda")
torch.manual_seed(0)
for wl in ("c3", "c4"):
    model = (pipeline.BaselineAnimator() if wl == "c3" else pipeline.SLRv1Animator()).to(dev).eval()
    image = torch.rand(1, 3, H, W, device=dev) * 2 - 1
    motion = torch.from_numpy(bench.smooth_motion(H, W)).to(dev)
    for frames in ([30], list(range(0, 60, 8)), list(range(60))):
        model.synthesize(image, motion, 60, frames=frames)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(3):
            model.synthesize(image, motion, 60, frames=frames)
        torch.cuda.synchronize()
        print(wl, len(frames), "frames:", round((time.perf_counter() - t0) / 3 * 1e3, 2), "ms")
