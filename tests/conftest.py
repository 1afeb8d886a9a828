import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


@pytest.fixture(scope="session")
def oracle():
    """The CPU oracle (test infrastructure; builds liboracle.so with gcc on first use)."""
    from oracle import oracle as o
    o.build()
    return o


def large_case(golden_dir, tag):
    """Inputs of the full-size digest fixtures (tests/golden/large_digests.npz), regenerated from
    the same seeds tools/make_golden.py used; only digests of the REFERENCE's outputs are stored."""
    import numpy as np
    g = np.load(f"{golden_dir}/large_digests.npz")
    _, C, H, W = [int(v) for v in g[f"{tag}_shape"]]
    r2 = np.random.default_rng(1000 + H)
    y, x = np.meshgrid(np.arange(H, dtype=np.float32), np.arange(W, dtype=np.float32), indexing="ij")
    u = 1.5 * np.sin(2 * np.pi * (2 * x / W + y / H) + 0.7)
    v = 1.5 * np.cos(2 * np.pi * (x / W - 1.5 * y / H) + 2.1)
    msk = (x >= 0.35 * W).astype(np.float32)
    motion = np.stack([u * msk, v * msk])[None].astype(np.float32)
    inp = r2.standard_normal((1, C, H, W)).astype(np.float32)
    return g, motion, inp, int(g[f"{tag}_steps"])


def a6_large_inputs(H=768, W=1280):
    """Seeded inputs of tests/golden/pipeline_a6_large.npz (same generator as tools/make_golden_pipeline.py
    ::a6_large_inputs; only digests of the REFERENCE's forward_flow outputs are stored)."""
    import numpy as np
    rng = np.random.default_rng(2000 + H)
    y, x = np.meshgrid(np.arange(H, dtype=np.float32), np.arange(W, dtype=np.float32), indexing="ij")
    u = 1.5 * np.sin(2 * np.pi * (2 * x / W + y / H) + 0.9)
    v = 1.5 * np.cos(2 * np.pi * (x / W - 1.5 * y / H) + 0.4)
    m = (x >= 0.35 * W).astype(np.float32)
    motion = np.stack([u * m, v * m])[None].astype(np.float32)
    fs = rng.standard_normal((1, 64, H, W)).astype(np.float32)
    Z = rng.standard_normal((1, 1, H, W)).astype(np.float32)
    alpha_out = rng.standard_normal((1, 2, H, W)).astype(np.float32)
    return fs, Z, motion, alpha_out


def v1_surface_inputs(W=60):
    """Seeded inputs of tests/golden/pipeline_v1_surface.npz (same generator as tools/make_golden_pipeline.py)."""
    import numpy as np
    rng = np.random.default_rng(4100 + W)
    y, x = np.meshgrid(np.arange(W, dtype=np.float32), np.arange(W, dtype=np.float32), indexing="ij")
    u = 1.2 * np.sin(2 * np.pi * (2 * x / W + y / W) + 0.2)
    v = 1.2 * np.cos(2 * np.pi * (x / W - 1.5 * y / W) + 1.3)
    motion = np.stack([u, v])[None].astype(np.float32)
    d = {"motion": motion,
         "fs": rng.standard_normal((1, 64, W, W)).astype(np.float32),
         "Z": (rng.standard_normal((1, 1, W, W)) * 2).astype(np.float32),
         "img": rng.uniform(-1, 1, (1, 3, W, W)).astype(np.float32),
         "alpha_out": rng.standard_normal((1, 2, W, W)).astype(np.float32),
         "bg_raw": rng.standard_normal((1, 3, W, W)).astype(np.float32),
         "dec_out": rng.standard_normal((1, 3, W, W)).astype(np.float32),
         "adec_out": (rng.standard_normal((1, 1, W, W)) * 2).astype(np.float32)}
    region = np.zeros((1, 1, W, W), np.float32)
    region[0, 0, W // 4: 3 * W // 4, W // 3:] = 1.0
    d["alpha_region"] = region
    return d
