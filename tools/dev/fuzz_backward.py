#!/usr/bin/env python
"""Fuzz of the splat backward (gradInput bit-exact, gradFlow to rounding) against the oracle through autograd of the drop-in operator:
ragged shapes, batches, flow families incl. bent rows (staged path), incoherent (direct path), far outside and non-finite vectors."""
import os, sys
import numpy as np, torch
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, ROOT)
import slr_sfs_amd as S
from oracle import oracle
oracle.build()
n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
bad = 0
for it in range(n_cases):
    N = int(rng.choice([1, 2])); C = int(rng.choice([1, 3, 4, 5, 17, 65])); H = int(rng.choice([1, 7, 8, 33, 64, 120])); W = int(rng.choice([1, 63, 64, 65, 200, 321]))
    yy, xx = np.meshgrid(np.arange(H, dtype=np.float32), np.arange(W, dtype=np.float32), indexing="ij")
    fam = int(rng.integers(0, 6))
    if fam == 0: fl = rng.uniform(-8, 8, (N, 2, H, W))
    elif fam == 1: fl = np.stack([np.stack([3 * np.sin(yy / 3 + k) + 0.02 * xx, 6 * np.sin(xx / 9 + k)]) for k in range(N)])      # bent rows
    elif fam == 2: fl = np.zeros((N, 2, H, W)) + rng.uniform(-1.5, 1.5, (N, 2, 1, 1))
    elif fam == 3: fl = rng.uniform(-2 * W, 2 * W, (N, 2, H, W))
    elif fam == 4: fl = np.stack([np.stack([(W / 2 - xx) * 0.7, (H / 2 - yy) * 0.7]) for _ in range(N)])
    else:
        fl = rng.uniform(-3, 3, (N, 2, H, W)); m = rng.random((N, 2, H, W)) < 0.01; fl[m] = rng.choice([np.nan, np.inf, -np.inf, 2e9], m.sum())
    fl = np.ascontiguousarray(fl, dtype=np.float32)
    x = rng.standard_normal((N, C, H, W)).astype(np.float32); go = rng.standard_normal((N, C, H, W)).astype(np.float32)
    xt = torch.from_numpy(x).cuda().requires_grad_(True); ft = torch.from_numpy(fl).cuda().requires_grad_(True)
    out = S.FunctionSoftsplat(xt, ft, None, "summation")
    out.backward(torch.from_numpy(go).cuda())
    gi_ref, gf_ref = oracle.softsplat_backward(x, fl, go)
    gi, gf = xt.grad.cpu().numpy(), ft.grad.cpu().numpy()
    ok_i = np.array_equal(gi, gi_ref)
    fin = np.isfinite(gf_ref)
    scale = max(1.0, float(np.abs(gf_ref[fin]).max()) if fin.any() else 1.0)
    err = float(np.abs(gf[fin] - gf_ref[fin]).max()) if fin.any() else 0.0
    if not ok_i or not (err <= 1e-5 * scale * max(1, C)):
        bad += 1
        print(f"MISMATCH case {it}: N{N} C{C} {H}x{W} family {fam}: gradInput equal {ok_i} (max diff {float(np.abs(gi - gi_ref).max()):.3e}) gradFlow err {err:.3e} scale {scale:.2e}", flush=True)
print(f"backward fuzz: {n_cases} cases, {bad} mismatches")
sys.exit(1 if bad else 0)
