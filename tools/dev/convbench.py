#!/usr/bin/env python
"""3x3 convolution of the decoder: split-f16 MFMA kernel (csrc/conv.hip) vs MIOpen fp32, per decoder
shape at the 768x1280 working resolution (development aid): error vs an fp64 convolution + time."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))); sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tools")); sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
import math
import os
import sys
import time

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import slr_sfs_amd as S  # noqa: F401
from slr_sfs_amd import _lib

L = _lib.lib()


def split_weights(w):
    cout, cin = w.shape[:2]
    amax = float(w.abs().max())
    wscale = 2.0 ** math.floor(math.log2(4096.0 / max(amax, 1e-30)))
    buf = torch.empty(L.slr_conv3x3_weight_bytes(cout, cin), dtype=torch.uint8, device=w.device)
    _lib.check(L.slr_conv3x3_split_weights(_lib.ptr(w), _lib.ptr(buf), cout, cin, wscale, _lib.stream_of(w)), "split")
    return buf, wscale


def conv_hip(x, buf, wscale, cout):
    n, cin, h, w = x.shape
    out = torch.empty(n, cout, h, w, device=x.device)
    _lib.check(L.slr_conv3x3_forward(_lib.ptr(x), _lib.ptr(buf), None, None, _lib.ptr(out), n, cin, cout, h, w, wscale,
                                     None, None, 0, _lib.stream_of(x)), "conv")
    return out


def timeit(fn, n=30):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


SHAPES = [(64, 64, 768, 1280), (64, 128, 768, 1280), (128, 128, 768, 1280), (128, 256, 384, 640),
          (256, 256, 384, 640), (256, 256, 192, 320), (256, 128, 192, 320), (128, 128, 384, 640)]
if len(sys.argv) > 1 and sys.argv[1] == "small":
    SHAPES = [(16, 64, 24, 40), (32, 128, 19, 45), (64, 64, 64, 96)]
torch.manual_seed(0)
tot_h = tot_m = 0.0
for cin, cout, h, w in SHAPES:
    x = torch.relu(torch.randn(1, cin, h, w, device="cuda")) * (torch.rand(1, 1, h, w, device="cuda") > 0.1)
    wt = torch.randn(cout, cin, 3, 3, device="cuda") * (1.0 / (3.0 * cin ** 0.5))
    buf, ws = split_weights(wt)
    y = conv_hip(x, buf, ws, cout)
    ym = F.conv2d(x, wt, None, 1, 1)
    hs = min(h, 96)
    ref = F.conv2d(x[:, :, :hs + 1].double(), wt.double(), None, 1, 1)[:, :, :hs]
    e_h = (y[:, :, :hs] - ref).abs().max().item()
    e_m = (ym[:, :, :hs] - ref).abs().max().item()
    t_h = timeit(lambda: conv_hip(x, buf, ws, cout))
    t_m = timeit(lambda: F.conv2d(x, wt, None, 1, 1))
    gf = 2.0 * 9 * cin * cout * h * w / 1e9
    tot_h += t_h
    tot_m += t_m
    print(f"{cin:4d}->{cout:4d} {h}x{w}: hip {t_h:7.3f} ms ({gf / t_h:6.1f} TF/s) err {e_h:.2e} | miopen {t_m:7.3f} ms "
          f"({gf / t_m:6.1f} TF/s) err {e_m:.2e} | ref max {ref.abs().max().item():.2f}", flush=True)
print(f"sum: hip {tot_h:.2f} ms, miopen {tot_m:.2f} ms")
