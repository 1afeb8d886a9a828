#!/usr/bin/env python
"""One front end (argv[1]: 1 scan, 2 rows) on config C2's grid with the smooth Euler t=30 flow (violent pile-ups at 256x480), 20 eager
calls: the workload for a rocprofv3 kernel trace of the front end's own kernels."""
import os, sys
import torch
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, ROOT)
import slr_sfs_amd as S
from bench import smooth_motion
dev = torch.device("cuda:0")
S._lib.lib().slr_splat_set_front_end(int(sys.argv[1]))
h, w = 256, 480
x, met = torch.randn(1, 64, h, w, device=dev), torch.randn(1, 1, h, w, device=dev)
fl = S.euler_integration(torch.from_numpy(smooth_motion(h, w)).to(dev), 30)[0]
for _ in range(20):
    S.FunctionSoftsplat(x, fl, met, "softmax")
torch.cuda.synchronize()
