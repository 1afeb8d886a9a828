#!/usr/bin/env python
"""Where rowbin_kernel's time goes (build: make -C slr-sfs_amd/csrc -B OUT=../lib/var_pstamp.so DEFS=-DSLR_PLAN_STAMPS):
100 MHz timestamps of the first workgroup's entry, the last workgroup's entry / arrival, and the plan's steps."""
import ctypes, os, sys
import numpy as np, torch
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, ROOT)
os.environ["SLR_SFS_AMD_LIB"] = os.path.join(ROOT, "slr-sfs_amd/lib/var_pstamp.so")
import slr_sfs_amd as S
L = S._lib.lib()
L.slr_debug_totals_offset.restype = ctypes.c_size_t
L.slr_debug_totals_offset.argtypes = [ctypes.c_int] * 4
L.slr_splat_set_front_end(2)
sys.path.insert(0, os.path.join(ROOT, "tools"))
from kbench import smooth_motion
for C, H, W, steps in ((65, 768, 1280, 0), (65, 768, 1280, 30), (65, 768, 1280, 59), (64, 256, 480, 0)):
    x = torch.randn(1, C, H, W, device="cuda"); out = torch.empty_like(x)
    fl = torch.zeros(1, 2, H, W, device="cuda") if steps == 0 else S.euler_integration(smooth_motion(H, W), steps)[0].contiguous()
    nb = L.slr_splat_workspace_bytes(1, C, H, W)
    ws = torch.zeros(nb, dtype=torch.uint8, device="cuda")
    off = L.slr_debug_totals_offset(1, C, H, W)
    for it in range(3):
        rc = L.slr_softsplat_forward(x.data_ptr(), fl.data_ptr(), out.data_ptr(), 1, C, H, W, ws.data_ptr(), nb, 0, None)
        assert rc == 0
        torch.cuda.synchronize()
        t = ws[off:off + 256].cpu().numpy().view(np.uint64).astype(np.int64)
        e0, eL, arr = t[15], t[14], t[13]
        p = t[8:13]
        print(f"{C}x{H}x{W} t={steps} run {it}: first WG entry 0 | last WG entry {(eL - e0) / 100:.2f} us | its arrival {(arr - e0) / 100:.2f} | plan entry {(p[0] - e0) / 100:.2f}"
              f" | loads {(p[1] - p[0]) / 100:.2f} | cut + scans {(p[3] - p[1]) / 100:.2f} | stores {(p[4] - p[3]) / 100:.2f} | end {(p[4] - e0) / 100:.2f}", flush=True)
