#!/usr/bin/env python
"""1x1 skip convolutions of the decoder: split-f16 kernel vs MIOpen (development aid)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))); sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tools")); sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
import os, sys, time
import torch, torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from slr_sfs_amd import nets


def timeit(fn, n=30):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e6


for cin, cout, h, w in [(64, 128, 768, 1280), (128, 256, 384, 640), (256, 128, 192, 320), (128, 128, 384, 640)]:
    conv = nets.Conv(cin, cout, 1, bias=False).cuda()
    x = torch.randn(1, cin, h, w, device="cuda")
    with torch.no_grad():
        t_h = timeit(lambda: conv(x))
        t_m = timeit(lambda: F.conv2d(x, conv.weight))
    mb = (cin + cout) * h * w * 4 / 1e6
    print(f"{cin}->{cout} {h}x{w}: hip {t_h:6.1f} us ({mb / t_h:5.2f} TB/s)  miopen {t_m:6.1f} us", flush=True)
