import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))); sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tools")); sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
import sys, ctypes, torch
pass
import slr_sfs_amd as S
from slr_sfs_amd import nets, _lib
L = _lib.lib()
torch.manual_seed(0)
def run(cin, cout, h, w, n=1):
    conv = nets.Conv(cin, cout, 3).cuda()
    x = torch.randn(n, cin, h, w, device="cuda")
    buf, ws = conv._split_weights()
    G = 1 << 16
    big = torch.full((G + n * cout * h * w + G,), 777.0, device="cuda")
    out = big[G:G + n * cout * h * w]
    for layout in (0, 2):
        if layout and cout % 8: continue
        big.fill_(777.0)
        _lib.check(L.slr_conv3x3_forward(_lib.ptr(x), _lib.ptr(buf), _lib.ptr(conv.bias), None, ctypes.c_void_p(out.data_ptr()),
                                         n, cin, cout, h, w, ws, None, None, layout, _lib.stream_of(x)), "conv")
        torch.cuda.synchronize()
        lo, hi = (big[:G] != 777.0).sum().item(), (big[G + out.numel():] != 777.0).sum().item()
        unwritten = (out == 777.0).sum().item()
        print(f"{cin}->{cout} {h}x{w} n={n} layout={layout}: guard writes before {lo} after {hi}; unwritten outputs {unwritten}", flush=True)
for args in [(64, 128, 40, 72), (64, 64, 40, 72), (64, 3, 40, 72), (128, 128, 20, 36), (64, 128, 40, 64), (3, 32, 40, 72), (64, 128, 7, 9, 2)]:
    run(*args)
