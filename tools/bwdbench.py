"""Timing of the backward kernels at 65x768x1280 (identity / Euler t=30 / t=59 flow):  python tools/bwdbench.py [flow name: only that one].
gradInput alone, gradFlow alone and both through slr_softsplat_backward_ws with the scratch it asks for (two channel groups: what autograd calls),
and both through slr_softsplat_backward (no scratch: one group)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import slr_sfs_amd as S
from kbench import timeit, smooth_motion
from slr_sfs_amd._lib import check, lib, ptr, stream_of
H, W, C = 768, 1280, 65
x = torch.randn(1, C, H, W, device="cuda"); go = torch.randn(1, C, H, W, device="cuda")
m = smooth_motion(H, W); dall, _ = S.euler_integration_all(m, 60)
gi = torch.empty_like(x); gf = torch.empty(1, 2, H, W, device="cuda")
L = lib(); st = stream_of(x)
nb = int(L.slr_softsplat_backward_ws_bytes(1, C, H, W)); ws = torch.empty(max(nb, 16), dtype=torch.uint8, device="cuda")
only = sys.argv[1] if len(sys.argv) > 1 else None
for name, fl in (("identity", torch.zeros(1, 2, H, W, device="cuda")), ("t30", dall[30:31].contiguous()), ("t59", dall[59:60].contiguous())):
    if only and name != only:
        continue
    t1 = timeit(lambda: check(L.slr_softsplat_backward_ws(ptr(x), ptr(fl), ptr(go), ptr(gi), None, 1, C, H, W, ptr(ws), nb, st), "b"), 10)
    t2 = timeit(lambda: check(L.slr_softsplat_backward_ws(ptr(x), ptr(fl), ptr(go), None, ptr(gf), 1, C, H, W, ptr(ws), nb, st), "b"), 10)
    t3 = timeit(lambda: check(L.slr_softsplat_backward_ws(ptr(x), ptr(fl), ptr(go), ptr(gi), ptr(gf), 1, C, H, W, ptr(ws), nb, st), "b"), 10)
    t4 = timeit(lambda: check(L.slr_softsplat_backward(ptr(x), ptr(fl), ptr(go), ptr(gi), ptr(gf), 1, C, H, W, st), "b"), 10)
    B = 2 * C * H * W * 4
    print(name, "grad_input us", t1, f"{B/t1[1]/1e6:.2f} TB/s", " grad_flow us", t2, f"{B/t2[1]/1e6:.2f} TB/s",
          " both (two groups + sum) us", t3, f"{1.5*B/t3[1]/1e6:.2f} TB/s", " both, no scratch (one group) us", t4)
