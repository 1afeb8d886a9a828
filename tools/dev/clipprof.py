#!/usr/bin/env python
"""Kernel breakdown of the per-clip fixed work (encoder + Euler passes) + one frame (development aid)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))); sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tools")); sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
from slr_sfs_amd import pipeline
from torch.profiler import profile, ProfilerActivity
H, W = 768, 1280
dev = torch.device("cuda")
torch.manual_seed(0)
model = pipeline.BaselineAnimator().to(dev).eval()
image = torch.rand(1, 3, H, W, device=dev) * 2 - 1
motion = torch.from_numpy(bench.smooth_motion(H, W)).to(dev)
for _ in range(2):
    model.synthesize(image, motion, 60, frames=[30])
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CUDA]) as prof:
    model.synthesize(image, motion, 60, frames=[30])
    torch.cuda.synchronize()
print(prof.key_averages().table(sort_by="cuda_time_total", row_limit=30, max_name_column_width=64))
