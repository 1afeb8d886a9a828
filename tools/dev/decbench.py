#!/usr/bin/env python
"""Decoder-only timing at 768x1280 (development aid): ms per pass + kernel breakdown."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))); sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tools")); sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import slr_sfs_amd as S

H, W = 768, 1280
torch.manual_seed(0)
dec = S.nets.DecoderPconv2(64, 3).cuda().eval()
x = torch.randn(1, 64, H, W, device="cuda")
x[:, :, 100:300, 200:500] = 0
if len(sys.argv) > 1 and sys.argv[1] == "cl":
    dec = dec.to(memory_format=torch.channels_last)
    x = x.contiguous(memory_format=torch.channels_last)
with torch.no_grad():
    for _ in range(3):
        y = dec(x)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        y = dec(x)
    torch.cuda.synchronize()
    print(f"decoder: {(time.perf_counter() - t0) / 10 * 1e3:.2f} ms/pass")
    from torch.profiler import profile, ProfilerActivity
    with profile(activities=[ProfilerActivity.CUDA]) as prof:
        for _ in range(3):
            y = dec(x)
        torch.cuda.synchronize()
    print(prof.key_averages().table(sort_by="cuda_time_total", row_limit=25, max_name_column_width=70))
