"""The 128->128 3x3 matrix-core convolution at 768x1280 back to back for argv[1] seconds; prints the TFLOP/s of every second."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))); sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tools")); sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from slr_sfs_amd import nets
secs = float(sys.argv[1]) if len(sys.argv) > 1 else 10
torch.manual_seed(0)
conv = nets.Conv(128, 128, 3).cuda()
x = torch.randn(1, 128, 768, 1280, device="cuda")
gf = 2.0 * 9 * 128 * 128 * 768 * 1280 / 1e12
with torch.no_grad():
    conv(x); torch.cuda.synchronize()
    t_end = time.time() + secs
    while time.time() < t_end:
        t0 = time.perf_counter()
        for _ in range(200):
            conv(x)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        print(f"{200 * gf / dt:.0f} TFLOP/s algorithmic ({dt / 200 * 1e6:.0f} us per launch)", flush=True)
