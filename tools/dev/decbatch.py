#!/usr/bin/env python
"""Decoder time per frame at batch 1 / 2 / 4 (development aid)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))); sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tools")); sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import slr_sfs_amd as S
H, W = 768, 1280
torch.manual_seed(0)
dec = S.nets.DecoderPconv2(64, 3).cuda().eval()
for n in (1, 2, 4, 1, 2):
    x = torch.randn(n, 64, H, W, device="cuda")
    x[:, :, 100:300, 200:500] = 0
    with torch.no_grad():
        for _ in range(3):
            dec(x)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(10):
            dec(x)
        torch.cuda.synchronize()
    print(f"batch {n}: {(time.perf_counter() - t0) / 10 / n * 1e3:.3f} ms per frame", flush=True)
