#!/usr/bin/env python
"""Channel-blocked variants of the matrix-core convolution at the decoder's heaviest shape (development aid)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))); sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tools")); sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from slr_sfs_amd import nets
I, O, R = nets.IN_B8, nets.OUT_B8, nets.RES_B8


def timeit(fn, n=30):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


torch.manual_seed(0)
for cin, cout, h, w in [(128, 128, 768, 1280), (256, 256, 384, 640)]:
    pc = nets.PartialConv(cin, cout, 3).cuda()
    x = torch.randn(1, cin, h, w, device="cuda")
    mask = (torch.rand(1, 1, h, w, device="cuda") > 0.2).float()
    sc, sh = torch.rand(cin, device="cuda") + 0.5, torch.randn(cin, device="cuda") * 0.3
    nb = (torch.rand(cout, device="cuda") + 0.5, torch.randn(cout, device="cuda") * 0.3)
    res = torch.randn(1, cout, h, w, device="cuda")
    gf = 2.0 * 9 * cin * cout * h * w / 1e9
    with torch.no_grad():
        rows = [("aa  NCHW in  -> NCHW out (pre+mask, next)", lambda: pc(x, mask, next_bn=nb, pre_bn=(sc, sh))),
                ("aa  B8 in    -> B8 out   (pre+mask, next)", lambda: pc(x, mask, next_bn=nb, pre_bn=(sc, sh), layout=I | O)),
                ("aa  NCHW in  -> B8 out   (pre+mask, next)", lambda: pc(x, mask, next_bn=nb, pre_bn=(sc, sh), layout=O)),
                ("ab  NCHW in  -> NCHW out (residual NCHW)", lambda: pc(x, mask, residual=res)),
                ("ab  B8 in    -> NCHW out (residual NCHW)", lambda: pc(x, mask, residual=res, layout=I)),
                ("ab  B8 in    -> B8 out   (residual B8)", lambda: pc(x, mask, residual=res, layout=I | O | R)),
                ("ab  B8 in    -> B8 out   (residual NCHW)", lambda: pc(x, mask, residual=res, layout=I | O)),
                ("ab  B8 in    -> B8 out   (no residual)", lambda: pc(x, mask, layout=I | O))]
        for name, fn in rows:
            t = timeit(fn)
            print(f"{cin}->{cout} {h}x{w} {name:44s} {t:7.3f} ms  {gf / t:6.1f} TF/s", flush=True)
