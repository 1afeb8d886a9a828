"""Per-workgroup lifetimes of the rows tile kernel (tracing build: make -C slr-sfs_amd/csrc -B OUT=../lib/var_trace.so DEFS=-DSLR_TRACE)."""
import ctypes, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
os.environ["SLR_SFS_AMD_LIB"] = os.path.join(ROOT, "slr-sfs_amd/lib/var_trace.so")
import slr_sfs_amd as S
from kbench import smooth_motion
L = S._lib.lib()
L.slr_debug_trace.argtypes = [ctypes.c_void_p]
L.slr_splat_set_front_end(2)
SL = 64
H, W = 768, 1280
xf = torch.randn(1, 65, H, W, device="cuda")
for name, n in (("t30", 30), ("t59", 59)):
    fl = S.euler_integration(smooth_motion(H, W), n)[0]
    S.FunctionSoftsplat(xf, fl, None, "summation")
    nb = 16384
    buf = torch.zeros(nb * SL, dtype=torch.int64, device="cuda")
    L.slr_debug_trace(buf.data_ptr())
    S.FunctionSoftsplat(xf, fl, None, "summation")
    torch.cuda.synchronize()
    L.slr_debug_trace(None)
    t = buf.cpu().numpy().reshape(nb, SL)
    t = t[(t[:, 48] > 0) & (t[:, 40] > 0)]
    life = (t[:, 40] - t[:, 41]) / 2200.0          # us (shader clock ~2.2 GHz)
    ent = t[:, 45]
    st = (t[:, 48] - t[:, 48].min()) / 100.0
    print(f"{name}: blocks {len(t)} life us p50 {np.median(life):.1f} p90 {np.percentile(life, 90):.1f} p99 {np.percentile(life, 99):.1f} max {life.max():.1f} | sum {life.sum():.0f} | last start {st.max():.1f} us")
    for lo, hi in ((0, 300), (300, 600), (600, 800), (800, 1100)):
        m = (ent >= lo) & (ent < hi)
        if m.any():
            print(f"   entries [{lo},{hi}): {m.sum():5d} blocks, life mean {life[m].mean():.1f} max {life[m].max():.1f}, max record list p50 {np.median(t[m, 43]):.0f} max {t[m, 43].max()}")
    order = np.argsort(-life)[:8]
    print("   longest:", [(round(float(life[i]), 1), int(ent[i]), int(t[i, 43])) for i in order])
