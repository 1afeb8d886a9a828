"""Phase timeline of the SCAN tile kernel's workgroups (development aid).  Needs the tracing build:
    make -C slr-sfs_amd/csrc -B OUT=../lib/var_trace.so DEFS=-DSLR_TRACE"""
import ctypes, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
os.environ["SLR_SFS_AMD_LIB"] = os.path.join(ROOT, "slr-sfs_amd/lib/var_trace.so")
import slr_sfs_amd as S
from kbench import smooth_motion
L = S._lib.lib()
L.slr_debug_trace.argtypes = [ctypes.c_void_p]
SL = 64
cases = []
x = torch.randn(1, 64, 256, 480, device="cuda"); met = torch.randn(1, 1, 256, 480, device="cuda")
cases.append(("c2 inc softmax", x, torch.rand(1, 2, 256, 480, device="cuda") * 16 - 8, met, "softmax"))
cases.append(("c2 id softmax", x, torch.zeros(1, 2, 256, 480, device="cuda"), met, "softmax"))
cases.append(("c2 id sum", x, torch.zeros(1, 2, 256, 480, device="cuda"), None, "summation"))
H, W = 768, 1280
xf = torch.randn(1, 65, H, W, device="cuda")
cases.append(("full id", xf, torch.zeros(1, 2, H, W, device="cuda"), None, "summation"))
cases.append(("full t30", xf, S.euler_integration(smooth_motion(H, W), 30)[0], None, "summation"))
cases.append(("full t59", xf, S.euler_integration(smooth_motion(H, W), 59)[0], None, "summation"))
if len(sys.argv) > 1:
    cases = [c for c in cases if any(a in c[0] for a in sys.argv[1:])]
for fe, thr in (("scan", 2**31 - 1), ("bins", 0)):
    L.slr_splat_set_scan_max_tiles(thr)
    for name, x, fl, met, mode in cases:
        S.FunctionSoftsplat(x, fl, met, mode)
        nb = 8192
        buf = torch.zeros(nb * SL, dtype=torch.int64, device="cuda")
        L.slr_debug_trace(buf.data_ptr())
        S.FunctionSoftsplat(x, fl, met, mode)
        torch.cuda.synchronize()
        L.slr_debug_trace(None)
        t = buf.cpu().numpy().reshape(nb, SL)
        t = t[t[:, 3] > 0]
        t0 = t[:, 0][t[:, 0] > 0].min() if fe == "bins" else t[:, 41].min()
        start = t[:, 41] if fe == "scan" else t[:, 0]
        med = lambda a: float(np.median(a))
        print(f"{fe:5s} {name:16s} blocks {len(t):5d} span {t[:, 40].max() - start.min():8d} ticks | start spread p50/p100 {med(start - start.min()):7.0f}/{(start - start.min()).max():7d}"
              f" | scan {med(t[:, 42] - t[:, 41]) if fe == 'scan' else 0:7.0f} (zero+bar {med(t[:, 33] - t[:, 41]):6.0f} boxes+bar {med(t[:, 34] - t[:, 33]):6.0f} cands {med(t[:, 35] - t[:, 34]):6.0f} bar {med(t[:, 42] - t[:, 35]):6.0f}) | 1a(ent+fp+atomics) {med(t[:, 1] - t[:, 0]):7.0f} (idx {med(t[:, 28] - t[:, 0]):6.0f})"
              f" | bar {med(t[:, 2] - t[:, 1]):6.0f} | scan+rec {med(t[:, 3] - t[:, 2]):6.0f} | ->stage0 {med(t[:, 4] - t[:, 3]):6.0f} | wait0 {med(t[:, 5] - t[:, 4]):6.0f}"
              f" | chunk0 {med(t[:, 6] - t[:, 5]):6.0f} chunk1 {med(t[:, 9] - t[:, 6]):6.0f} chunk2 {med(t[:, 12] - t[:, 9]):6.0f} | total {med(t[:, 40] - start):8.0f} p99 {np.percentile(t[:, 40] - start, 99):8.0f}"
              f" | entries p50 {med(t[:, 45]):6.0f} max {t[:, 45].max()}" + (f" | first loads issue->landed {med(t[:, 29] - t[:, 28]):6.0f}" if (t[:, 29] > 0).any() else ""))
        if fe == "scan":
            full = buf.cpu().numpy().reshape(nb, SL)
            ok = (full[:, 48] > 0) & (full[:, 50] > 0)
            t = full[ok]
            life = t[:, 38] - t[:, 41]; own = t[:, 37] - t[:, 41]; pops = t[:, 39]
            r0 = t[:, 48].min()
            st, oe, ex = (t[:, 48] - r0) / 100.0, (t[:, 49] - r0) / 100.0, (t[:, 50] - r0) / 100.0       # us
            k1 = ex.max()
            edges = np.linspace(0, k1, 21)
            alive = [int(((st < e1) & (ex > e0)).sum()) for e0, e1 in zip(edges[:-1], edges[1:])]
            print(f"      span {k1:.1f} us; last start {st.max():.1f} us, last own end {oe.max():.1f} us; life us p50/p99/max {np.percentile(ex - st, 50):.1f}/{np.percentile(ex - st, 99):.1f}/{(ex - st).max():.1f}")
            print("      alive per 5% of the span:", alive)
            LF = ex - st
            for nm, sel in (("all", pops >= 0), ("no pieces", pops == 0), ("with pieces", pops > 0)):
                if sel.any():
                    print(f"      life us [{nm}: {int(sel.sum())}] mean {LF[sel].mean():.1f} sum {LF[sel].sum():.0f} pct 50/75/90/95/99 " + "/".join(f"{np.percentile(LF[sel], q):.0f}" for q in (50, 75, 90, 95, 99))
                          + f" | own-work us mean {(oe - st)[sel].mean():.1f} | after own work us mean {(ex - oe)[sel].mean():.1f}")
            first = st < 2.0                       # cold start (kernel arguments, instruction cache) vs later rounds
            for nm, sel in (("first round", first), ("later", ~first)):
                if sel.any():
                    print(f"      {nm:12s} [{int(sel.sum())}] zero+bar {np.median((t[:, 33] - t[:, 41])[sel]):6.0f} boxes {np.median((t[:, 34] - t[:, 33])[sel]):6.0f} cands {np.median((t[:, 35] - t[:, 34])[sel]):6.0f}"
                          f" 1a {np.median((t[:, 1] - t[:, 0])[sel]):6.0f} rec {np.median((t[:, 3] - t[:, 2])[sel]):6.0f} ->stage0 {np.median((t[:, 4] - t[:, 3])[sel]):6.0f}")
            heavy = t[:, 36] > 0
            print(f"      kernel span {t[:, 38].max() - t[:, 41].min():8d} | WG life p50/p99/max {np.percentile(life, 50):8.0f}/{np.percentile(life, 99):8.0f}/{life.max():8d}"
                  f" | own work p50/p99/max {np.percentile(own, 50):8.0f}/{np.percentile(own, 99):8.0f}/{own.max():8d} | heavy homes {int(heavy.sum())}: own work p50 {np.median(own[heavy]) if heavy.any() else 0:8.0f}"
                  f" | WGs that popped {int((pops > 0).sum())}, pops max {pops.max()}, total {pops.sum()} | pop check (end - own end) p50 {np.median(t[:, 38][pops == 0] - t[:, 37][pops == 0]):6.0f}"
                  f" | last own-work end {t[:, 37].max() - t[:, 41].min():8d}, last start {t[:, 41].max() - t[:, 41].min():8d}")
