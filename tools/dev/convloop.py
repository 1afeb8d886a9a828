#!/usr/bin/env python
"""The decoder's 128->128 3x3 partial convolution at 768x1280 back to back for argv[1] seconds (argv[2] = fp32: on the fp32 rung):
the load tools/dev/power_watch.sh samples rocm-smi next to."""
import os, sys, time
import torch
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, ROOT)
import slr_sfs_amd  # noqa: F401
from slr_sfs_amd import nets
secs = float(sys.argv[1]) if len(sys.argv) > 1 else 10.0
f32 = len(sys.argv) > 2 and sys.argv[2] == "fp32"
H, W, C = 768, 1280, 128
dev = torch.device("cuda:0")
pc = nets.PartialConv(C, C, 3).to(dev)
x = torch.randn(1, C, H, W, device=dev)
mask = (torch.rand(1, 1, H, W, device=dev) > 0.1).float()
sc, sh = torch.rand(C, device=dev) + 0.5, torch.randn(C, device=dev) * 0.3
nb = (torch.rand(C, device=dev) + 0.5, torch.randn(C, device=dev) * 0.3)
lay = nets.IN_B8 | nets.OUT_B8
flops = 2.0 * 9 * C * C * H * W
ctx = nets.fp32_kernels() if f32 else torch.no_grad()
with torch.no_grad(), ctx:
    t_end = time.time() + secs
    while time.time() < t_end:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(200):
            pc(x, mask, next_bn=nb, pre_bn=(sc, sh), layout=lay)
        e1.record()
        e1.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / 200
        print(f"convloop{' fp32' if f32 else ''}: {flops / us / 1e6:.0f} TFLOP/s algorithmic ({us:.0f} us per launch)", flush=True)
