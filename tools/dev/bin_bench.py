#!/usr/bin/env python
"""slr_splat_bin alone (rowbin_kernel + plan) and the whole one-flow call at 65 x 768 x 1280, graph-replayed: us per call."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.argv = sys.argv[:1] + ["none"]
import frontend_bench as F
from slr_sfs_amd._lib import lib, check, ptr, stream_of, workspace
S = F.S
C, H, W = 65, 768, 1280
x = torch.randn(1, C, H, W, device="cuda"); out = torch.empty_like(x)
m = torch.from_numpy(F.smooth_motion(H, W)).cuda()
flows = {"id": torch.zeros(1, 2, H, W, device="cuda"), "t30": S.euler_integration(m, 30)[0], "t59": S.euler_integration(m, 59)[0],
         "inc": torch.rand(1, 2, H, W, device="cuda") * 16 - 8}
L = lib(); ws = workspace(x, "a", 1, C, H, W)
for name, fl in flows.items():
    fl = fl.contiguous()
    def binonly():
        check(L.slr_splat_bin(ptr(fl), 1, H, W, ptr(ws), ws.numel(), stream_of(x)), "b")
    def full():
        check(L.slr_softsplat_forward(ptr(x), ptr(fl), ptr(out), 1, C, H, W, ptr(ws), ws.numel(), 0, stream_of(x)), "f")
    ref = None
    full(); torch.cuda.synchronize(); d = out.double().sum().item()
    print(f"{name:4s} bin {F.graph_us(binonly):6.1f}  call {F.graph_us(full):6.1f} us   checksum {d:.6f}", flush=True)
