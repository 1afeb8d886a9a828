#!/usr/bin/env python
"""Sweep of slr_splat_set_scan_shape (column pieces x channel groups in the first launch; workgroups x groups of the pass-by-pass launch)
on the small grids of bench.py's roofline_dropin: graph-replayed call time per shape, result checked against the rows front end."""
import os, sys
import torch
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, ROOT)
import slr_sfs_amd as S
from bench import smooth_motion, _graph_call_us
dev = torch.device("cuda:0")
L = S._lib.lib()
shapes = [tuple(int(v) for v in a.split(",")) for a in sys.argv[1:]] or [(0, 0, 0, 0), (1, 2, 0, 0), (2, 1, 0, 0), (2, 2, 0, 0), (4, 1, 0, 0), (4, 2, 0, 0), (8, 1, 0, 0),
                                                                         (4, 1, 32, 2), (4, 1, 32, 4), (4, 1, 64, 4), (1, 2, 32, 4)]
torch.manual_seed(0)
for c, h, w in ((64, 256, 480), (64, 128, 240), (65, 384, 640)):
    x, met = torch.randn(1, c, h, w, device=dev), torch.randn(1, 1, h, w, device=dev)
    for fname, fl in (("inc", torch.rand(1, 2, h, w, device=dev) * 16 - 8), ("t30", S.euler_integration(torch.from_numpy(smooth_motion(h, w)).to(dev), 30)[0])):
        L.slr_splat_set_front_end(2)
        ref = S.FunctionSoftsplat(x, fl, met, "softmax").clone()
        L.slr_splat_set_front_end(1)
        row = []
        for sh in shapes:
            L.slr_splat_set_scan_shape(*sh)
            out = S.FunctionSoftsplat(x, fl, met, "softmax")
            err = float((out - ref).abs().max())
            us = _graph_call_us(lambda: S.FunctionSoftsplat(x, fl, met, "softmax"))
            row.append(f"{sh[0]}x{sh[1]}/{sh[2]}x{sh[3]}: {us:6.1f}" + ("" if err < 2e-4 else f" ERR {err:.2e}"))
        L.slr_splat_set_scan_shape(0, 0, 0, 0)
        L.slr_splat_set_front_end(0)
        print(f"{c}x{h}x{w} {fname}: " + " | ".join(row), flush=True)
