#!/usr/bin/env python
"""Config C2's grid under flows of different coherence (scan front end vs bins): where do the 38 us go?"""
import os, sys
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import slr_sfs_amd as S
from bench import smooth_motion
from frontend_bench import measure
dev = torch.device("cuda:0")
C, h, w = 64, 256, 480
x, met = torch.randn(1, C, h, w, device=dev), torch.randn(1, 1, h, w, device=dev)
alg = (2 * C + 3) * h * w * 4
mo = torch.from_numpy(smooth_motion(h, w)).to(dev)
for tag, fl in (("identity", torch.zeros(1, 2, h, w, device=dev)), ("t3", S.euler_integration(mo, 3)[0]),
                ("U(-1,1)", torch.rand(1, 2, h, w, device=dev) * 2 - 1), ("U(-4,4)", torch.rand(1, 2, h, w, device=dev) * 8 - 4),
                ("U(-8,8)", torch.rand(1, 2, h, w, device=dev) * 16 - 8)):
    measure(f"C2 {tag} softmax", x, fl, met, "softmax", alg)
    measure(f"C2 {tag} sum", x, fl, None, "summation", alg - h * w * 4)
