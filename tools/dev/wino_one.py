"""One layer of the fp32 rung for PMC passes: python tools/dev/wino_one.py [direct] [cin cout h w]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import slr_sfs_amd as S
from slr_sfs_amd import nets
args = [a for a in sys.argv[1:] if a != "direct"]
cin, cout, h, w = (int(v) for v in args) if len(args) == 4 else (128, 128, 768, 1280)
conv = nets.Conv(cin, cout, 3).cuda()
x = torch.randn(1, cin, h, w, device="cuda")
with torch.no_grad(), nets.fp32_kernels(winograd="direct" not in sys.argv[1:]):
    for _ in range(6):
        y = conv(x)
torch.cuda.synchronize()
print("ok", float(y.abs().mean()))
