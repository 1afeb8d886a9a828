#!/usr/bin/env python
"""tests/golden/euler_batch.npz: the reference's EulerIntegration module (models/projection/euler_integration_manipulator.py:58-71)
run UNMODIFIED on a batch of 16 motion fields with per-sample step counts, and torch autograd's gradient through it w.r.t. the
motion fields (the training step's use, animating_softmax_splating.py:579-580).  Needs /root/reference (see tools/make_golden.py:
the same import of the reference; only numeric arrays are written).
Usage:  python tools/make_golden_euler_batch.py"""
import os
import shutil
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import make_golden as mg                                                          # noqa: E402


def main():
    _, eim = mg.load_reference()
    rng = np.random.default_rng(606)
    out = {}
    for tag, (H, W) in {"a": (16, 24), "b": (33, 47)}.items():
        fields = []
        for rep in range(4):
            f = mg.motion_fields(H, W, rng)
            fields += [f["smooth"] * (1.0 + rep), f["random3"], f["halfint"], f["exit"]]
        motion = np.concatenate(fields, 0).astype(np.float32)                     # [16,2,H,W]
        steps = np.array([0, 1, 2, 5, 17, 60, 3, 7, 0, 31, 1, 12, 59, 4, 9, 2], np.int64)
        m = torch.from_numpy(motion).clone().requires_grad_(True)
        disp, vis = eim.EulerIntegration()(m, torch.from_numpy(steps), show_visible_pixels=True)
        gout = torch.from_numpy(rng.standard_normal(motion.shape).astype(np.float32))
        (gm,) = torch.autograd.grad(disp, m, gout)
        out[f"{tag}_motion"], out[f"{tag}_steps"] = motion, steps
        out[f"{tag}_disp"], out[f"{tag}_vis"] = disp.detach().numpy().astype(np.float32), vis.numpy().astype(np.float32)
        out[f"{tag}_gout"], out[f"{tag}_gmotion"] = gout.numpy(), gm.numpy().astype(np.float32)
    path = os.path.join(mg.OUT, "euler_batch.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path) // 1024, "kB")


if __name__ == "__main__":
    try:
        main()
    finally:
        shutil.rmtree(mg.TMP, ignore_errors=True)
