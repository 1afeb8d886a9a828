#!/usr/bin/env python
"""Time-to-phase of the scanning tile kernel at config C2 (measurement builds -DSLR_CUT=k end the kernel at phase k):
    for k in 1 2 3 4 5; do make -C slr-sfs_amd/csrc -B OUT=../lib/var_cut$k.so DEFS=-DSLR_CUT=$k; done
    tools/dev/variants.sh "python tools/dev/c2_cuts.py" cut1 cut2 cut3 cut4 cut5 default"""
import os, sys
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import slr_sfs_amd as S
from frontend_bench import graph_us
dev = torch.device("cuda:0")
shapes = [(64, 256, 480)] + ([(65, 768, 1280)] if "full" in sys.argv else [])
for C, h, w in shapes:
    x, met = torch.randn(1, C, h, w, device=dev), torch.randn(1, 1, h, w, device=dev)
    S._lib.lib().slr_splat_set_front_end(2 if "rows" in sys.argv else 1)
    out = []
    from bench import smooth_motion
    mo = torch.from_numpy(smooth_motion(h, w)).to(dev)
    for tag, fl in (("id", torch.zeros(1, 2, h, w, device=dev)), ("inc", torch.rand(1, 2, h, w, device=dev) * 16 - 8),
                    ("t30", S.euler_integration(mo, 30)[0]), ("t59", S.euler_integration(mo, 59)[0])):
        for mode, m in (("sum", None),):
            out.append(f"{tag}/{mode} {graph_us(lambda: S.FunctionSoftsplat(x, fl, m, mode if m is not None else 'summation')):6.1f}")
    print(f"{C}x{h}x{w}: " + " | ".join(out), flush=True)
