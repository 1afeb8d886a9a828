"""Seeded inputs of tests/golden/config_literal.npz (BASELINE.json configs C1 / C2 as stated; tools/make_golden_configs.py
ran the reference's own operators on them).  Test infrastructure only."""
import zlib

import numpy as np


def _rng(*what):
    return np.random.default_rng(zlib.crc32("/".join(str(w) for w in what).encode()))


def smooth_motion(H, W, amp=1.5):
    y, x = np.meshgrid(np.arange(H, dtype=np.float32), np.arange(W, dtype=np.float32), indexing="ij")
    u = amp * np.sin(2 * np.pi * (2 * x / W + y / H) + 0.3)
    v = amp * np.cos(2 * np.pi * (x / W - 1.5 * y / H) + 1.1)
    m = (x >= 0.35 * W).astype(np.float32)
    return np.stack([u * m, v * m])[None].astype(np.float32)


def config_inputs(tag):
    """-> (input [1,C,H,W], metric [1,1,H,W], motion or None, Euler steps, flow or None)."""
    H, W = 256, 480
    r = _rng("config", tag)
    if tag == "c1":                                         # one frame: an image in [-1, 1], N = 5
        return (r.uniform(-1, 1, (1, 3, H, W)).astype(np.float32), r.standard_normal((1, 1, H, W)).astype(np.float32),
                smooth_motion(H, W), 5, None)
    x = r.standard_normal((1, 64, H, W)).astype(np.float32)
    metric = r.standard_normal((1, 1, H, W)).astype(np.float32)
    if tag == "c2_inc":
        return x, metric, None, 0, r.uniform(-8, 8, (1, 2, H, W)).astype(np.float32)
    assert tag == "c2_smooth"
    return x, metric, smooth_motion(H, W), 30, None


def digest_positions(tag, size, n=4096):
    return _rng("config-digest", tag, size).integers(0, size, n)
