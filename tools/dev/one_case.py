#!/usr/bin/env python
"""One small-grid case of the one-flow operator, 30 calls (for rocprofv3 --kernel-trace --stats): python tools/dev/one_case.py [inc|t30|t59] [n h w c] [mode]"""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import slr_sfs_amd as S
from bench import smooth_motion
dev = torch.device("cuda:0")
which = sys.argv[1] if len(sys.argv) > 1 else "t30"
n, h, w, c = (int(v) for v in sys.argv[2:6]) if len(sys.argv) > 5 else (1, 256, 480, 64)
mode = sys.argv[6] if len(sys.argv) > 6 else "softmax"
torch.manual_seed(0)
x, met = torch.randn(n, c, h, w, device=dev), torch.randn(n, 1, h, w, device=dev)
if which == "inc":
    fl = torch.rand(n, 2, h, w, device=dev) * 16 - 8
else:
    mo = torch.from_numpy(np.concatenate([smooth_motion(h, w, seed=i) for i in range(n)], 0)).to(dev)
    fl = S.EulerIntegration()(mo, torch.full((n,), int(which[1:]), device=dev)).contiguous()
for _ in range(30):
    S.FunctionSoftsplat(x, fl, None if mode == "summation" else met, mode)
torch.cuda.synchronize()
