#!/usr/bin/env python
"""Fuzz of the clip path (MotionPlan + fused two-flow kernel, one frame and batches; baseline and 2-layer model) against the oracle:
random ragged shapes, motion families (smooth, strong pile-ups, static regions, a few non-finite vectors) and weight ranges."""
import os, sys
import numpy as np, torch
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, ROOT)
import slr_sfs_amd as S
from slr_sfs_amd import synthesis
from oracle import oracle
oracle.build()
n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
bad = 0
for it in range(n_cases):
    C = int(rng.choice([1, 3, 8, 17, 64])); H = int(rng.choice([8, 9, 40, 64, 100, 136])); W = int(rng.choice([64, 65, 100, 192, 250]))
    N = int(rng.choice([2, 5, 9, 16]))
    yy, xx = np.meshgrid(np.arange(H, dtype=np.float32), np.arange(W, dtype=np.float32), indexing="ij")
    fam = int(rng.integers(0, 5))
    amp = float(rng.choice([0.5, 1.5, 4.0]))
    if fam == 0: mo = np.stack([amp * np.sin(xx / 11 + yy / 23), amp * np.cos(yy / 9 - xx / 31)])
    elif fam == 1: mo = np.stack([amp * np.sign(W / 2 - xx) * np.minimum(np.abs(W / 2 - xx) / 6, 1), 0.5 * amp * np.sign(H / 2 - yy)])   # converging: pile-ups
    elif fam == 2: mo = np.stack([amp * np.sin(xx / 7), amp * np.cos(yy / 5)]) * (xx > W / 3)                                     # static third
    elif fam == 3: mo = rng.uniform(-amp, amp, (2, H, W))
    else:
        mo = np.stack([amp * np.sin(xx / 13), amp * np.sin(yy / 17)]); m = rng.random((2, H, W)) < 0.003; mo[m] = rng.choice([np.nan, np.inf, 1e9], m.sum())
    mo = np.ascontiguousarray(mo[None], dtype=np.float32)
    fs = rng.standard_normal((1, C, H, W)).astype(np.float32)
    Z = (rng.standard_normal((1, 1, H, W)) * float(rng.choice([0.5, 3.0, 20.0]))).astype(np.float32)
    v1 = bool(rng.integers(0, 2))
    d = lambda a: torch.from_numpy(a).cuda()
    if v1:
        af = rng.standard_normal((1, 1, H, W)).astype(np.float32) * 2; abg = rng.uniform(0, 1, (1, 1, H, W)).astype(np.float32)
        cs = synthesis.ClipSynthesizer(d(fs), d(Z), d(mo), N, alpha_fluid_logit=d(af), alpha_bg=d(abg))
    else:
        cs = synthesis.ClipSynthesizer(d(fs), d(Z), d(mo), N)
    ts = sorted(set(int(t) for t in rng.integers(0, N, 3)))
    out = torch.empty(len(ts), C, H, W, device="cuda"); oa = torch.empty(len(ts), 1, H, W, device="cuda") if v1 else None
    cs.features_batch(ts, out, oa)
    for k, t in enumerate(ts):
        one = cs.features(t)
        if v1:
            g_ref, a_ref, _ = oracle.synth_v1(fs, Z, af, abg, mo, t, N)
            got = [(out[k:k + 1], g_ref), (oa[k:k + 1], a_ref), (one[0], g_ref), (one[1], a_ref)]
        else:
            g_ref = oracle.synth_baseline(fs, Z, mo, t, N)
            got = [(out[k:k + 1], g_ref), (one if torch.is_tensor(one) else one[0], g_ref)]
        for j, (g, r) in enumerate(got):
            g = g.cpu().numpy()
            fin = np.isfinite(r)
            scale = max(1.0, float(np.abs(r[fin]).max()) if fin.any() else 1.0)
            err = float(np.abs(g[fin] - r[fin]).max()) if fin.any() else 0.0
            if not (err <= 2e-4 * scale) or not np.array_equal(np.isfinite(g), fin):
                bad += 1
                print(f"MISMATCH case {it} C{C} {H}x{W} N{N} fam {fam} amp {amp} v1 {v1} t {t} out {j}: err {err:.3e} scale {scale:.2e}", flush=True)
print(f"clip fuzz: {n_cases} cases, {bad} mismatches")
sys.exit(1 if bad else 0)
