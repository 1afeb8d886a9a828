#!/usr/bin/env python
"""Kernel breakdown of the per-frame work of the C3 pipeline (development aid)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))); sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tools")); sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
from slr_sfs_amd import pipeline
from torch.profiler import profile, ProfilerActivity
H, W = 768, 1280
dev = torch.device("cuda")
torch.manual_seed(0)
wl = sys.argv[1] if len(sys.argv) > 1 else "c3"
model = (pipeline.BaselineAnimator() if wl == "c3" else pipeline.SLRv1Animator()).to(dev).eval()
image = torch.rand(1, 3, H, W, device=dev) * 2 - 1
motion = torch.from_numpy(bench.smooth_motion(H, W)).to(dev)
frames = list(range(5, 60, 5))
model.synthesize(image, motion, 60, frames=frames)
torch.cuda.synchronize()
t0 = time.perf_counter()
model.synthesize(image, motion, 60, frames=frames)
torch.cuda.synchronize()
print(f"{len(frames)} frames: {(time.perf_counter() - t0) * 1e3:.1f} ms")
with profile(activities=[ProfilerActivity.CUDA]) as prof:
    model.synthesize(image, motion, 60, frames=frames)
    torch.cuda.synchronize()
print(prof.key_averages().table(sort_by="cuda_time_total", row_limit=28, max_name_column_width=64))
