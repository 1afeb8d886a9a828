"""Counters of the scan front end (development aid).  Needs: make -C slr-sfs_amd/csrc OUT=../lib/var_stats.so DEFS=-DSLR_SCAN_STATS"""
import ctypes, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
os.environ["SLR_SFS_AMD_LIB"] = os.path.join(ROOT, "slr-sfs_amd/lib/var_stats.so")
import slr_sfs_amd as S
from slr_sfs_amd import _lib
from kbench import smooth_motion
L = _lib.lib()
L.slr_debug_ctl_offset.restype = ctypes.c_size_t
L.slr_debug_ctl_offset.argtypes = [ctypes.c_int] * 4
L.slr_splat_set_scan_max_tiles(2**31 - 1)
H, W, C = 768, 1280, 65
x = torch.randn(1, C, H, W, device="cuda")
m = smooth_motion(H, W)
for name, steps in (("t30", 30), ("t59", 59)):
    fl = S.euler_integration(m, steps)[0]
    S.FunctionSoftsplat(x, fl, None, "summation")
    torch.cuda.synchronize()
    ws = _lib.workspace(x, "a", 1, C, H, W)
    off = L.slr_debug_ctl_offset(1, C, H, W)
    c = ws[off:off + 256].view(torch.int32).cpu().numpy()
    print(name, "head/tail/slots", c[0], c[1], c[2], "| scans m0/m1/m2", c[8], c[9], c[10], "| candidates per scan m0/m1/m2",
          c[12] / max(1, c[8]), c[13] / max(1, c[9]), c[14] / max(1, c[10]), "| heavy tiles", c[16], "segments", c[17], "max ns", c[18],
          "not shared", c[19], "| item spins", c[20], "| claims", c[22])
