#!/usr/bin/env python
"""Config C2 (FunctionSoftsplat softmax, 64 channels, 256x480, incoherent U(-8,8) flow) and two more small grids:
the tile kernel alone (HIP events recorded by the library around that launch) and the whole call.
Development aid (SLR_SFS_AMD_LIB=variant)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import slr_sfs_amd as S
from slr_sfs_amd import synthesis
from kbench import timeit
dev = torch.device("cuda:0")
out = []
grids = ((64, 256, 480), (64, 128, 240), (65, 384, 640))
for C, h, w in (grids[:1] if sys.argv[1:] == ["c2"] else grids):        # "c2": that grid only (for rocprofv3 traces)
    x, met = torch.randn(1, C, h, w, device=dev), torch.randn(1, 1, h, w, device=dev)
    fl = torch.rand(1, 2, h, w, device=dev) * 16 - 8
    alg = (2 * C + 3) * h * w * 4
    synthesis.kernel_timing = []
    for _ in range(45):
        synthesis._arm_timer(x)
        S.FunctionSoftsplat(x, fl, met, "softmax")
    torch.cuda.synchronize()
    us = sorted(a.elapsed_time(b) * 1e3 for a, b, _ in synthesis.kernel_timing[5:])
    synthesis.kernel_timing = None
    k = sum(us) / len(us)
    call = timeit(lambda: S.FunctionSoftsplat(x, fl, met, "softmax"), 40)[0]
    out.append(f"{C}x{h}x{w}: tile {k:5.1f} us ({alg / k / 1e3 / 8000:.3f})  call {call:5.1f} us ({alg / call / 1e3 / 8000:.3f})")
print(os.path.basename(os.environ.get("SLR_SFS_AMD_LIB", "default")), " | ".join(out))
