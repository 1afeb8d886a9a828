"""Phase timeline of the tile kernel's workgroups (development aid).  Needs the tracing build:
    make -C slr-sfs_amd/csrc -B OUT=../lib/var_trace.so DEFS=-DSLR_TRACE
(one-flow kernel only: the stamp table has room for its 9 chunks)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))); sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tools")); sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
import ctypes, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ["SLR_SFS_AMD_LIB"] = os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "slr-sfs_amd/lib/var_trace.so")
import slr_sfs_amd as S
from kbench import smooth_motion
L = S._lib.lib()
L.slr_debug_trace.argtypes = [ctypes.c_void_p]
H, W, C = 768, 1280, 65
x = torch.randn(1, C, H, W, device="cuda")
m = smooth_motion(H, W)
dall, _ = S.euler_integration_all(m, 60)
for name, fl in (("identity", torch.zeros(1, 2, H, W, device="cuda")), ("t30", dall[30:31].contiguous()), ("t59", dall[59:60].contiguous())):
    S.FunctionSoftsplat(x, fl, None, "summation")
    nb = 4096
    buf = torch.zeros(nb * 40, dtype=torch.int64, device="cuda")
    L.slr_debug_trace(buf.data_ptr())
    S.FunctionSoftsplat(x, fl, None, "summation")
    torch.cuda.synchronize()
    L.slr_debug_trace(None)
    t = buf.cpu().numpy().reshape(nb, 40)
    act = t[:, 3] > 0
    t = t[act]
    t0 = t[:, 0].min()
    print(name, "active blocks", len(t), "kernel span (ticks)", (t[:, 4:31].max() - t0))
    d = lambda a, b: np.median(t[:, b] - t[:, a])
    print("  phase1a:", d(0, 1), " barrier:", d(1, 2), " scan+records:", d(2, 3), " -> first stage done:", d(3, 4))
    for c in range(9):
        b = 4 + 3 * c
        print(f"  chunk {c}: barrier wait {np.median(t[:, b+1]-t[:, b]):8.0f}  gather+store {np.median(t[:, b+2]-t[:, b+1]):8.0f}" +
              (f"  barrier+stage {np.median(t[:, b+3]-t[:, b+2]):8.0f}" if c < 8 else ""))
    tot = t[:, 30] - t[:, 0]
    print("  sum of block totals / 1e6:", tot.sum() / 1e6, " blocks slower than 1.5x median:", int((tot > 1.5 * np.median(tot)).sum()),
          " their share of the sum:", float(tot[tot > 1.5 * np.median(tot)].sum() / tot.sum()))
    print("  block total percentiles 50/90/99/max:", np.percentile(tot, [50, 90, 99, 100]))
    idx = np.argsort(tot)[-8:]
    for i in idx:
        print(f"    slow block: total {tot[i]:8d}  max records/pixel {t[i,33]:6d}  wave0 recs {t[i,34]:6d}  bin count {t[i,35]:6d} seg {t[i,36]}  phase1 {t[i,3]-t[i,0]:7d}")
    print("  corr(total, maxrec) =", np.corrcoef(tot, t[:, 33])[0, 1], " corr(total, bincount) =", np.corrcoef(tot, np.minimum(t[:, 35], 1024))[0, 1])
    print("  block total median", np.median(t[:, 30] - t[:, 0]), "max", (t[:, 30] - t[:, 0]).max(), "start spread", np.percentile(t[:, 0] - t0, [0, 25, 50, 75, 100]))
