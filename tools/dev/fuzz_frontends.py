#!/usr/bin/env python
"""Fuzz of the one-flow front ends: random shapes (ragged edges, tiny images, batches), flow families (smooth, incoherent, collapsing,
far outside, non-finite sprinkles) and modes (the four FunctionSoftsplat modes + the maximum splat); the scan and the rows front end
against the CPU oracle on the same inputs (cases of up to 2 M elements; the larger ones against each other)."""
import os, sys
import numpy as np, torch
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, ROOT)
import slr_sfs_amd as S
from oracle import oracle
oracle.build()
L = S._lib.lib()
n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 200
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
bad = 0
for it in range(n_cases):
    N = int(rng.choice([1, 1, 2, 3])); C = int(rng.choice([1, 3, 8, 17, 33, 65]))
    H = int(rng.choice([1, 5, 8, 9, 31, 64, 100, 200, 256])); W = int(rng.choice([1, 7, 63, 64, 65, 130, 200, 300, 480]))
    yy, xx = np.meshgrid(np.arange(H, dtype=np.float32), np.arange(W, dtype=np.float32), indexing="ij")
    if it % 25 == 24:                                                   # now and then a big one
        N, C, H, W = int(rng.choice([1, 2])), int(rng.choice([9, 65])), int(rng.choice([384, 768])), int(rng.choice([640, 1280]))
        yy, xx = np.meshgrid(np.arange(H, dtype=np.float32), np.arange(W, dtype=np.float32), indexing="ij")
    fam = rng.integers(0, 10)
    if fam == 0: fl = rng.uniform(-8, 8, (N, 2, H, W))
    elif fam == 1: fl = np.stack([np.stack([3 * np.sin(xx / 17 + k) + 0.3 * yy / max(H, 1), 2 * np.cos(yy / 11 + k)]) for k in range(N)])
    elif fam == 2: fl = np.stack([np.stack([(W * rng.uniform(0, 1) - xx) * rng.uniform(0.5, 1.0), (H * rng.uniform(0, 1) - yy) * rng.uniform(0.5, 1.0)]) for _ in range(N)])
    elif fam == 3: fl = rng.uniform(-3 * W, 3 * W, (N, 2, H, W))
    elif fam == 4: fl = np.zeros((N, 2, H, W)) + rng.uniform(-1.5, 1.5, (N, 2, 1, 1))
    elif fam == 5: fl = np.stack([np.stack([(xx % 5) - xx + W // 2, (yy % 3) - yy + H // 2]) for _ in range(N)])        # onto a 5 x 3 patch
    elif fam == 7: fl = np.stack([np.stack([W * rng.uniform(0.1, 0.9) - xx + (yy % 2), 0.3 * np.sin(yy / 9) + 0 * xx]) for _ in range(N)])   # onto a 2-pixel column
    elif fam == 8: fl = np.stack([np.stack([1.7 * np.cos(xx / 13) + 0 * yy, H * rng.uniform(0.1, 0.9) - yy + (xx % 3)]) for _ in range(N)])   # onto 3 rows
    elif fam == 9: fl = np.stack([np.stack([(xx - W / 2) * rng.uniform(-0.9, -0.3), (yy - H / 2) * rng.uniform(-0.4, 0.4)]) for _ in range(N)])  # squeeze / stretch
    else:
        fl = rng.uniform(-4, 4, (N, 2, H, W)); m = rng.random((N, 2, H, W)) < 0.01
        fl[m] = rng.choice([np.nan, np.inf, -np.inf, 3e9, -3e9], m.sum())
    fl = torch.from_numpy(np.ascontiguousarray(fl, dtype=np.float32)).cuda()
    x = torch.randn(N, C, H, W, device="cuda"); met = torch.randn(N, 1, H, W, device="cuda") * 0.7
    mode = ["summation", "average", "linear", "softmax", "maximum"][int(rng.integers(0, 5))]
    m_ = None if mode in ("summation", "average", "maximum") else (met.abs() + 0.1 if mode == "linear" else met)
    outs = {}
    for fe in (1, 2):
        prev = L.slr_splat_set_front_end(fe)
        outs[fe] = S.ModuleMaximumsplat()(x, fl) if mode == "maximum" else S.FunctionSoftsplat(x, fl, m_, mode)
        L.slr_splat_set_front_end(prev)
    if N * C * H * W <= 2_000_000:                          # the CPU oracle is the reference (non-finite flows: same finite pattern)
        xn, fn = x.cpu().numpy(), fl.cpu().numpy()
        refn = oracle.maxsplat_forward(xn, fn, 0.0) if mode == "maximum" else oracle.function_softsplat(xn, fn, None if m_ is None else m_.cpu().numpy(), mode)
        ref = torch.from_numpy(np.ascontiguousarray(refn)).cuda()
        which = (1, 2)
    else:
        ref, which = outs[1], (2,)
    scale = max(1.0, float(ref[torch.isfinite(ref)].abs().max()) if bool(torch.isfinite(ref).any()) else 1.0)
    for fe in which:
        fin = torch.isfinite(ref) & torch.isfinite(outs[fe])
        err = float((outs[fe][fin] - ref[fin]).abs().max()) if bool(fin.any()) else 0.0
        same_fin = bool(torch.equal(torch.isfinite(outs[fe]), torch.isfinite(ref)))
        if not (err <= 3e-4 * scale) or not same_fin:
            bad += 1
            print(f"MISMATCH case {it} fe {fe}: N{N} C{C} {H}x{W} family {fam} mode {mode} err {err:.3e} scale {scale:.2e} finite-equal {same_fin}", flush=True)
print(f"fuzz: {n_cases} cases, {bad} mismatches")
sys.exit(1 if bad else 0)
