#!/usr/bin/env python
"""The splat stage of bench.py's workload alone (no networks): both Euler passes + 60 x
(bin + fused two-direction splat + normalise) at 768x1280 -- the command the PMC passes
(FETCH_SIZE / WRITE_SIZE) are collected on (rocprofv3 --pmc crashes with the MIOpen pipeline)."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import slr_sfs_amd as S
from bench import smooth_motion, H, W, NFRAMES
dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(0)
fs = torch.randn(1, 64, H, W, device=dev, generator=g)
Z = torch.randn(1, 1, H, W, device=dev, generator=g)
motion = torch.from_numpy(smooth_motion(H, W)).to(dev)
cs = S.synthesis.ClipSynthesizer(fs, Z, motion, NFRAMES)
for rep in range(int(sys.argv[1]) if len(sys.argv) > 1 else 1):
    for t in range(NFRAMES):
        out = cs.features(t)
torch.cuda.synchronize()
print("ok", float(out.abs().mean()))
