"""Timing of pathological flows (development aid): everything converging to a point / row / column."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))); sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tools")); sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import slr_sfs_amd as S
from kbench import timeit
H, W, C = 768, 1280, 65
x = torch.randn(1, C, H, W, device="cuda")
yy, xx = torch.meshgrid(torch.arange(H, device="cuda", dtype=torch.float32), torch.arange(W, device="cuda", dtype=torch.float32), indexing="ij")
z = torch.zeros_like(xx)
cases = {
    "point": torch.stack([(W / 2 - xx) * 0.999 + 0.3, (H / 2 - yy) * 0.999 - 0.2])[None],
    "row": torch.stack([z, (H / 2 - yy) * 0.999 - 0.2])[None],
    "column": torch.stack([(W / 2 - xx) * 0.999 + 0.3, z])[None],
    "shrink4x": torch.stack([(W / 2 - xx) * 0.75, (H / 2 - yy) * 0.75])[None],
}
for name, fl in cases.items():
    fl = fl.contiguous()
    t = timeit(lambda: S.FunctionSoftsplat(x, fl, None, "summation"), 5, 2)
    print(name, "full call us (median, min):", t)
