"""Backward at 65x768x1280 through slr_softsplat_backward_ws (channel groups when the library asks for scratch): identity / Euler t=30 / t=59."""
import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import slr_sfs_amd as S
from kbench import timeit, smooth_motion
from slr_sfs_amd._lib import check, lib, ptr, stream_of
H, W, C = 768, 1280, 65
x = torch.randn(1, C, H, W, device="cuda"); go = torch.randn(1, C, H, W, device="cuda")
m = smooth_motion(H, W); dall, _ = S.euler_integration_all(m, 60)
gi = torch.empty_like(x); gf = torch.empty(1, 2, H, W, device="cuda")
L = lib(); st = stream_of(x)
nb = int(L.slr_softsplat_backward_ws_bytes(1, C, H, W)); ws = torch.empty(max(nb, 16), dtype=torch.uint8, device="cuda")
print("ws bytes", nb)
for name, fl in (("identity", torch.zeros(1, 2, H, W, device="cuda")), ("t30", dall[30:31].contiguous()), ("t59", dall[59:60].contiguous())):
    t1 = timeit(lambda: check(L.slr_softsplat_backward_ws(ptr(x), ptr(fl), ptr(go), ptr(gi), None, 1, C, H, W, ptr(ws), nb, st), "b"), 10)
    t3 = timeit(lambda: check(L.slr_softsplat_backward_ws(ptr(x), ptr(fl), ptr(go), ptr(gi), ptr(gf), 1, C, H, W, ptr(ws), nb, st), "b"), 10)
    print(name, "grad_input us", t1, " both us", t3)
