#!/usr/bin/env python
"""Tile kernel of the drop-in one-flow operator alone (HIP events recorded by the library around that launch):
C = 65, 768x1280, Euler t=30 / t=59 / incoherent flows.  Development aid for knob sweeps (SLR_SFS_AMD_LIB=variant)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import slr_sfs_amd as S
from slr_sfs_amd import synthesis
from bench import smooth_motion, H, W
from kbench import timeit
dev = torch.device("cuda:0")
motion = torch.from_numpy(smooth_motion(H, W)).to(dev)
x = torch.randn(1, 65, H, W, device=dev)
alg = (2 * 65 + 2) * H * W * 4
flows = {"id": torch.zeros(1, 2, H, W, device=dev), "t30": S.euler_integration(motion, 30)[0], "t59": S.euler_integration(motion, 59)[0],
         "inc": torch.rand(1, 2, H, W, device=dev) * 16 - 8}
out = []
for name, fl in flows.items():
    synthesis.kernel_timing = []
    for _ in range(25):
        synthesis._arm_timer(x)
        S.FunctionSoftsplat(x, fl, None, "summation")
    torch.cuda.synchronize()
    us = sorted(a.elapsed_time(b) * 1e3 for a, b, _ in synthesis.kernel_timing[5:])
    synthesis.kernel_timing = None
    call = timeit(lambda: S.FunctionSoftsplat(x, fl, None, "summation"), 20)[0]
    out.append(f"{name} {sum(us)/len(us):6.1f} us ({alg/ (sum(us)/len(us)) / 1e3 / 8000:.3f}) call {call:6.1f}")
print(os.path.basename(os.environ.get("SLR_SFS_AMD_LIB", "default")), " | ".join(out))
