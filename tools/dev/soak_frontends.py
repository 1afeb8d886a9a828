"""Soak of the one-flow front ends (the scan front end shares heavy tiles between workgroups through a queue in the
workspace: look for rare hand-over races).  Repeats calls on flows with many heavy tiles and compares every result with the
first one of its front end (bit-for-bit is not expected across front ends: summation order) and with the bins result within
rounding; also alternates shapes and streams on one workspace cache."""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import slr_sfs_amd as S
from kbench import smooth_motion
L = S._lib.lib()
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 200
fe_under_test = int(sys.argv[2]) if len(sys.argv) > 2 else 1            # 1 scan, 2 rows
torch.manual_seed(0)
cases = []
for (C, H, W, steps, amp) in ((65, 768, 1280, 59, 1.5), (64, 256, 480, 30, 1.5), (16, 128, 240, 30, 1.5), (7, 200, 328, 40, 3.0)):
    m = smooth_motion(H, W, amp=amp)
    fl = S.euler_integration(m, steps)[0]
    x = torch.randn(1, C, H, W, device="cuda")
    met = torch.randn(1, 1, H, W, device="cuda") * 0.5
    cases.append((x, fl, met))
bad = 0
t0 = time.time()
for ci, (x, fl, met) in enumerate(cases):
    L.slr_splat_set_front_end(3 - fe_under_test)          # the OTHER front end is the reference (1 scan <-> 2 rows)
    ref_sum = S.FunctionSoftsplat(x, fl, None, "summation")
    ref_soft = S.FunctionSoftsplat(x, fl, met, "softmax")
    L.slr_splat_set_front_end(fe_under_test)
    scale = float(ref_sum.abs().max())
    for r in range(reps):
        a = S.FunctionSoftsplat(x, fl, None, "summation")
        b = S.FunctionSoftsplat(x, fl, met, "softmax")
        e1 = float((a - ref_sum).abs().max()) / scale
        e2 = float((b - ref_soft).abs().max())
        if not (e1 < 2e-5 and e2 < 2e-4 and bool(torch.isfinite(a).all())):
            bad += 1
            print("MISMATCH case", ci, "rep", r, e1, e2, flush=True)
    print("case", ci, tuple(x.shape), "ok", flush=True)
# two streams, each with its own workspaces, interleaved
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
x, fl, met = cases[1]
ref = S.FunctionSoftsplat(x, fl, met, "softmax")
torch.cuda.synchronize()
for r in range(reps):
    with torch.cuda.stream(s1):
        a = S.FunctionSoftsplat(x, fl, met, "softmax")
    with torch.cuda.stream(s2):
        b = S.FunctionSoftsplat(x, fl, met, "softmax")
    torch.cuda.synchronize()
    if float((a - ref).abs().max()) > 2e-4 or float((b - ref).abs().max()) > 2e-4:
        bad += 1
        print("MISMATCH streams rep", r, flush=True)
print("soak done:", bad, "mismatches,", round(time.time() - t0, 1), "s")
sys.exit(1 if bad else 0)
