#!/usr/bin/env python
"""The rows front end on the 768x1280 flows (graph time per call), for A/B runs over library variants (tools/dev/variants.sh)."""
import os, sys
import torch
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import slr_sfs_amd as S
from bench import smooth_motion
from frontend_bench import graph_us
dev = torch.device("cuda:0")
S._lib.lib().slr_splat_set_front_end(int(os.environ.get("FE", "2")))
H, W = 768, 1280
motion = torch.from_numpy(smooth_motion(H, W)).to(dev)
x = torch.randn(1, 65, H, W, device=dev)
flows = {"id": torch.zeros(1, 2, H, W, device=dev), "t15": S.euler_integration(motion, 15)[0], "t30": S.euler_integration(motion, 30)[0],
         "t45": S.euler_integration(motion, 45)[0], "t59": S.euler_integration(motion, 59)[0], "inc": torch.rand(1, 2, H, W, device=dev) * 16 - 8}
out = []
for name, fl in flows.items():
    ts = sorted(graph_us(lambda: S.FunctionSoftsplat(x, fl, None, "summation")) for _ in range(3))
    out.append(f"{name} {ts[1]:6.1f}")
print(" | ".join(out), flush=True)
