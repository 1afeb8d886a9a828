#!/usr/bin/env python
"""One front end of the one-flow operator (argv[1]: 0 bins, 1 scan, 2 rows) on the 768x1280 flows, 20 eager calls each:
the workload for a rocprofv3 kernel trace of the front end's own kernels."""
import os, sys
import torch
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import slr_sfs_amd as S
from bench import smooth_motion
dev = torch.device("cuda:0")
S._lib.lib().slr_splat_set_front_end(int(sys.argv[1]))
which = sys.argv[2:] or ["id", "t30", "t59", "inc", "c2"]
H, W = 768, 1280
motion = torch.from_numpy(smooth_motion(H, W)).to(dev)
x = torch.randn(1, 65, H, W, device=dev)
flows = {"id": torch.zeros(1, 2, H, W, device=dev), "t30": S.euler_integration(motion, 30)[0],
         "t59": S.euler_integration(motion, 59)[0], "inc": torch.rand(1, 2, H, W, device=dev) * 16 - 8}
for name in which:
    if name == "c2":
        xx, fl, met = torch.randn(1, 64, 256, 480, device=dev), torch.rand(1, 2, 256, 480, device=dev) * 16 - 8, torch.randn(1, 1, 256, 480, device=dev)
        f = lambda: S.FunctionSoftsplat(xx, fl, met, "softmax")
    else:
        fl = flows[name]
        f = lambda: S.FunctionSoftsplat(x, fl, None, "summation")
    for _ in range(20):
        f()
    torch.cuda.synchronize()
