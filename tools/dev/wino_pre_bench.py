"""The decoder's heaviest layer form (BN + mask prologue, partial epilogue + next BN, channel-blocked) on the three rungs.  python tools/dev/wino_pre_bench.py"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
dev = torch.device("cuda:0")
for _ in range(2):
    d = bench.conv_roofline(dev, fp32=True)
    print(os.path.basename(os.environ.get("SLR_SFS_AMD_LIB", "default")), "direct", d["avg_us"], "winograd", d["winograd"]["avg_us"], d["winograd"]["min_us"])
