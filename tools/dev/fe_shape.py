#!/usr/bin/env python
"""20 eager calls of the one-flow operator on one shape with a smooth Euler flow: the workload for a rocprofv3 kernel trace.
usage: python tools/dev/fe_shape.py <n> <c> <h> <w> <softmax|summation> <t>"""
import os, sys
import numpy as np, torch
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, ROOT)
import slr_sfs_amd as S
from bench import smooth_motion
dev = torch.device("cuda:0")
n, c, h, w = (int(v) for v in sys.argv[1:5])
mode, t = sys.argv[5], int(sys.argv[6])
x, met = torch.randn(n, c, h, w, device=dev), torch.randn(n, 1, h, w, device=dev)
mo = torch.from_numpy(np.concatenate([smooth_motion(h, w, seed=i) for i in range(n)], 0)).to(dev)
fl = S.EulerIntegration()(mo, torch.full((n,), t, device=dev)).contiguous()
with torch.no_grad():
    for _ in range(20):
        S.FunctionSoftsplat(x, fl, met if mode != "summation" else None, mode)
torch.cuda.synchronize()
