"""fp32 rung of the 3x3 convolution (v_mfma_f32_32x32x2_f32; direct and Winograd F(2x2,3x3)) vs the split-f16 rung and MIOpen fp32: time and TFLOP/s per layer shape."""
import os, sys, time
import torch, torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import slr_sfs_amd as S
from slr_sfs_amd import nets
dev = torch.device("cuda:0")
LAY = 0 if '--nchw' in sys.argv else nets.IN_B8 | nets.OUT_B8        # the networks' activations are channel-blocked ([N,C/8,H,W,8])

def timeit(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3

for cin, cout, h, w in ((128, 128, 768, 1280), (256, 256, 384, 640), (64, 128, 768, 1280), (64, 64, 768, 1280), (128, 3, 768, 1280)):
    if cout % 8 and LAY: continue
    conv = nets.Conv(cin, cout, 3).to(dev)
    x = torch.randn(1, cin, h, w, device=dev)
    fl = 2.0 * cin * cout * 9 * h * w
    with torch.no_grad():
        t_split = timeit(lambda: conv(x, layout=LAY))
        with nets.fp32_kernels(winograd=False):
            t_f32 = timeit(lambda: conv(x, layout=LAY))
        with nets.fp32_kernels(winograd=True):
            t_w = timeit(lambda: conv(x, layout=LAY))
        t_mi = timeit(lambda: F.conv2d(x, conv.weight, conv.bias, padding=1), 5)
    print(f"{cin:4d}->{cout:4d} {h}x{w}: split-f16 {t_split:8.1f} us ({fl / t_split / 1e6:6.1f} TF) | fp32 rung {t_f32:8.1f} us ({fl / t_f32 / 1e6:6.1f} TF = {fl / t_f32 / 1e6 / 157.3:.2f} of 157.3) | Winograd fp32 {t_w:8.1f} us ({fl / t_w / 1e6:6.1f} TF direct-equivalent) | MIOpen fp32 {t_mi:8.1f} us ({fl / t_mi / 1e6:6.1f} TF)")
