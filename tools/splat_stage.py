#!/usr/bin/env python
"""The splat stage of bench.py's workload alone (no networks): both Euler passes + 60 x
(fused two-direction splat + normalise, 8 frames per launch) at 768x1280, bins and plans per clip -- the command the PMC passes
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
V1 = "v1" in sys.argv[1:]          # the 2-layer model's packing: + the alpha group (own weights) per frame
if V1:
    af = torch.randn(1, 1, H, W, device=dev, generator=g)
    abg = torch.sigmoid(torch.randn(1, 1, H, W, device=dev, generator=g))
    cs = S.synthesis.ClipSynthesizer(fs, Z, motion, NFRAMES, alpha_fluid_logit=af, alpha_bg=abg)
else:
    cs = S.synthesis.ClipSynthesizer(fs, Z, motion, NFRAMES)
B = S.synthesis.MAX_BATCH                     # frames per launch of the tile kernel, as the pipelines use it
out = torch.empty(B, 64, H, W, device=dev)
outa = torch.empty(B, 1, H, W, device=dev) if V1 else None
reps = [int(a) for a in sys.argv[1:] if a.isdigit()]
for rep in range(reps[0] if reps else 1):
    for t0 in range(0, NFRAMES, B):
        ts = list(range(t0, min(t0 + B, NFRAMES)))
        cs.features_batch(ts, out[:len(ts)], None if outa is None else outa[:len(ts)])
torch.cuda.synchronize()
print("ok", float(out.abs().mean()))
