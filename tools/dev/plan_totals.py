#!/usr/bin/env python
"""Plan totals (items, partial slots, heavy items) of the bins and rows front ends on the 768x1280 flows
(build: make -C slr-sfs_amd/csrc -B OUT=../lib/var_pstamp.so DEFS=-DSLR_PLAN_STAMPS)."""
import ctypes, os, sys
import numpy as np, torch
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, ROOT)
os.environ["SLR_SFS_AMD_LIB"] = os.path.join(ROOT, "slr-sfs_amd/lib/var_pstamp.so")
import slr_sfs_amd as S
from bench import smooth_motion
L = S._lib.lib()
L.slr_debug_totals_offset.restype = ctypes.c_size_t
L.slr_debug_totals_offset.argtypes = [ctypes.c_int] * 4
C, H, W = 65, 768, 1280
mo = torch.from_numpy(smooth_motion(H, W)).cuda()
x = torch.randn(1, C, H, W, device="cuda"); out = torch.empty_like(x)
nb = L.slr_splat_workspace_bytes(1, C, H, W)
ws = torch.zeros(nb, dtype=torch.uint8, device="cuda")
off = L.slr_debug_totals_offset(1, C, H, W)
L.slr_debug_rowcnt_offset.restype = ctypes.c_size_t
L.slr_debug_rowcnt_offset.argtypes = [ctypes.c_int] * 4
roff = L.slr_debug_rowcnt_offset(1, C, H, W)
nt = (H // 8) * (W // 64)
for name, fl in (("id", torch.zeros(1, 2, H, W, device="cuda")), ("t30", S.euler_integration(mo, 30)[0]), ("t59", S.euler_integration(mo, 59)[0]),
                 ("inc", torch.rand(1, 2, H, W, device="cuda") * 16 - 8)):
    for fe in (0, 2):
        L.slr_splat_set_front_end(fe)
        assert L.slr_softsplat_forward(x.data_ptr(), fl.data_ptr(), out.data_ptr(), 1, C, H, W, ws.data_ptr(), nb, 0, None) == 0
        torch.cuda.synchronize()
        t = ws[off:off + 32].cpu().numpy().view(np.uint32)
        if fe == 2:
            rc3 = ws[roff:roff + nt * 16].cpu().numpy().view(np.uint64).reshape(nt, 2)
            rc = rc3[:, 0]
            octs = 16 * np.stack([(rc3[:, 1] >> np.uint64(8 * o)) & np.uint64(0xff) for o in range(8)], 1).astype(np.int64)
            rows, ent = (rc & np.uint64(0xffffffff)).astype(np.int64), (rc >> np.uint64(32)).astype(np.int64)
            hv = ent > 1024
            print(f"     octant sums / entries (heavy tiles): {octs[hv].sum() / max(ent[hv].sum(), 1):.2f}; max octant {octs.max()}; heavy tiles {hv.sum()}")
            print(f"     rows per tile mean {rows.mean():.1f} p50 {np.median(rows):.0f} p99 {np.percentile(rows, 99):.0f} max {rows.max()} | entries mean {ent.mean():.0f} max {ent.max()} | scanned px / entries {64 * rows.sum() / ent.sum():.2f}"
                  f" | heavy tiles (>877): rows mean {rows[ent > 877].mean() if (ent > 877).any() else 0:.0f}, entries mean {ent[ent > 877].mean() if (ent > 877).any() else 0:.0f}")
        print(f"{name:4s} fe {fe}: items {t[0]} partial slots {t[1]} multi {t[3]} whole {t[4]} heavy-first items {t[5]}")
