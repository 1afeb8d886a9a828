"""Time of the split-f16 3x3 kernels against the number of 16-channel input chunks (fixed cost per tile vs cost per chunk):
   python tools/dev/conv_k_sweep.py [H W]   -- channel-blocked in / out, BN + ReLU prologue, N = 4"""
import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from slr_sfs_amd import nets
H, W = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (768, 1280)
N = 4
dev = torch.device("cuda:0")
def timeit(fn, n=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(True), torch.cuda.Event(True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
with torch.no_grad():
    for cout in (64, 128):
        for cin in (32, 64, 128, 256):
            conv = nets.Conv(cin, cout, 3).to(dev)
            x = torch.randn(N, cin, H, W, device=dev)
            sc, sh = torch.rand(cin, device=dev) + 0.5, torch.randn(cin, device=dev) * 0.3
            us_pre = timeit(lambda: conv(x, (sc, sh), layout=nets.IN_B8 | nets.OUT_B8))
            us_raw = timeit(lambda: conv(x, None, layout=nets.IN_B8 | nets.OUT_B8))
            macs = N * H * W * cin * cout * 9
            print(f"cout {cout:4d} cin {cin:4d} chunks {cin // 16:3d}: prologue {us_pre:8.1f} us  plain {us_raw:8.1f} us   "
                  f"{2 * macs / us_raw / 1e6:7.1f} TFLOP/s algorithmic (plain)", flush=True)
