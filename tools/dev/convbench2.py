#!/usr/bin/env python
"""Cost of the fused prologue / epilogue of the matrix-core convolution (development aid)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))); sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tools")); sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
import os, sys, time
import torch
import torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import slr_sfs_amd as S  # noqa
from slr_sfs_amd import nets


def timeit(fn, n=30):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


torch.manual_seed(0)
for cin, cout, h, w in [(128, 128, 768, 1280), (64, 128, 768, 1280), (256, 256, 384, 640)]:
    pc = nets.PartialConv(cin, cout, 3).cuda()
    x = torch.randn(1, cin, h, w, device="cuda")
    mask = (torch.rand(1, 1, h, w, device="cuda") > 0.2).float()
    sc, sh = torch.rand(cin, device="cuda") + 0.5, torch.randn(cin, device="cuda") * 0.3
    nsc, nsh = torch.rand(cout, device="cuda") + 0.5, torch.randn(cout, device="cuda") * 0.3
    res = torch.randn(1, cout, h, w, device="cuda")
    gf = 2.0 * 9 * cin * cout * h * w / 1e9
    with torch.no_grad():
        rows = [("plain conv", lambda: nets.Conv.conv(pc, x, None)),
                ("conv + bias", lambda: nets.Conv.conv(pc, x, pc.bias)),
                ("pre-BN conv", lambda: nets.Conv.conv(pc, x, None, (sc, sh))),
                ("pconv chain (no pre) + next", lambda: pc(x, mask, next_bn=(nsc, nsh))),
                ("pconv chain (no pre) + residual", lambda: pc(x, mask, residual=res)),
                ("pconv pre+mask + next", lambda: pc(x, mask, next_bn=(nsc, nsh), pre_bn=(sc, sh))),
                ("pconv pre+derived + next", lambda: pc(x, None, next_bn=(nsc, nsh), pre_bn=(sc, sh)))]
        for name, fn in rows:
            t = timeit(fn)
            print(f"{cin}->{cout} {h}x{w} {name:34s} {t:7.3f} ms  {gf / t:6.1f} TF/s", flush=True)
