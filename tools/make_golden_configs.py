#!/usr/bin/env python
"""tests/golden/config_literal.npz: BASELINE.json's configs C1 and C2 as they are stated, run through the REFERENCE's own
operators (models/softsplat.py FunctionSoftsplat with its kernel text compiled for the host, tools/make_golden.py;
models/projection/euler_integration_manipulator.py), digests only:
  c1        configs[0]: one 1x3x256x480 frame, motion integrated over N = 5 steps, softmax splat with a metric plane
  c2_inc    configs[1]: random 64-channel 256x480 feature + incoherent U(-8,8) flow, softmax
  c2_smooth configs[1] with a smooth motion field integrated to t = 30
Inputs are regenerated from seeds by the tests (config_inputs below is imported by them).  Needs /root/reference."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
from config_inputs import config_inputs, digest_positions  # noqa: E402


def main():
    import torch
    import make_golden as MG
    ss, eim = MG.load_reference()
    g = {}
    for tag in ("c1", "c2_inc", "c2_smooth"):
        x, metric, motion, steps, flow = config_inputs(tag)
        if flow is None:
            flow = eim.euler_integration(torch.from_numpy(motion), steps)[0].numpy().astype(np.float32)
            g[f"{tag}_flow_sum"] = flow.astype(np.float64).sum(axis=(2, 3))
        out = ss.FunctionSoftsplat(MG.cudalike(x), MG.cudalike(flow), MG.cudalike(metric), "softmax")
        out = out.detach().as_subclass(torch.Tensor).numpy().astype(np.float32)
        pos = digest_positions(tag, out.size)
        g[f"{tag}_shape"] = np.array(out.shape, np.int64)
        g[f"{tag}_val"] = out.ravel()[pos]
        g[f"{tag}_plane_sums"] = out.astype(np.float64).sum(axis=(2, 3))
        g[f"{tag}_holes"] = np.int64((out == 0).all(axis=1).sum())
        g[f"{tag}_absmax"] = np.float32(np.abs(out).max())
        print(tag, out.shape, "absmax", float(np.abs(out).max()), "holes", int(g[f"{tag}_holes"]))
    p = os.path.join(ROOT, "tests", "golden", "config_literal.npz")
    np.savez_compressed(p, **g)
    print("wrote", p, os.path.getsize(p) // 1024, "kB")


if __name__ == "__main__":
    import shutil
    try:
        main()
    finally:
        import make_golden as MG
        shutil.rmtree(MG.TMP, ignore_errors=True)
