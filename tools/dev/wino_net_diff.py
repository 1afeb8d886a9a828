"""Frames of the v1 / baseline animators on the fp32 rung: Winograd vs direct 3x3 (where and how much they differ).  python tools/dev/wino_net_diff.py"""
import os, sys, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_large_golden as T
from slr_sfs_amd import nets
gd = os.path.join(ROOT, "tests", "golden")
g = np.load(f"{gd}/native_frames_768.npz")
S, N = int(g["S"]), int(g["N"])
img, motion, _ = T.NF.e2e_inputs(S, N)
img, motion = torch.from_numpy(img).cuda(), torch.from_numpy(motion).cuda()
keys = ("PredImg", "FluidImg", "CompositeFluidAlpha")
v1 = T._v1(gd).cuda(); v1.convs = "fp32"
base = T._baseline(gd).cuda(); base.convs = "fp32"
res = {}
for wino in (True, False):
    v1.convs = base.convs = "fp32-winograd" if wino else "fp32"
    res[wino] = (v1.synthesize(img, motion, N, frames=[30], keys=keys), base.synthesize(img, motion, N, frames=[30]))
for k in keys:
    d = (res[True][0][k] - res[False][0][k]).abs()
    print(f"v1 {k}: max |wino - direct| {float(d.max()):.3e}, mean {float(d.mean()):.3e}, pixels > 2e-5: {int((d > 2e-5).sum())} of {d.numel()}, > 1e-4: {int((d > 1e-4).sum())}")
    if float(d.max()) > 2e-5:
        idx = torch.nonzero(d > 0.5 * d.max())[:8]
        print("   at", idx.tolist())
d = (res[True][1] - res[False][1]).abs()
print(f"baseline PredImg: max {float(d.max()):.3e}, mean {float(d.mean()):.3e}, pixels > 2e-5: {int((d > 2e-5).sum())}")
