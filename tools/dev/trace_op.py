"""Where a workgroup of the one-flow rows kernel spends its life (tracing build: make -C slr-sfs_amd/csrc OUT=../lib/var_trace.so DEFS=-DSLR_TRACE).
usage: python tools/dev/trace_op.py [t30|t59|id|inc]"""
import ctypes, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
os.environ.setdefault("SLR_SFS_AMD_LIB", os.path.join(ROOT, "slr-sfs_amd/lib/var_trace.so"))
import slr_sfs_amd as S
from bench import smooth_motion, H, W
L = S._lib.lib()
L.slr_debug_trace.argtypes = [ctypes.c_void_p]
L.slr_splat_set_front_end(2)
dev = torch.device("cuda:0")
which = sys.argv[1] if len(sys.argv) > 1 else "t30"
x = torch.randn(1, 65, H, W, device=dev)
motion = torch.from_numpy(smooth_motion(H, W)).to(dev)
fl = {"id": lambda: torch.zeros(1, 2, H, W, device=dev), "t30": lambda: S.euler_integration(motion, 30)[0], "t59": lambda: S.euler_integration(motion, 59)[0],
      "inc": lambda: torch.rand(1, 2, H, W, device=dev) * 16 - 8}[which]()
S.FunctionSoftsplat(x, fl, None, "summation")
nb, SL = 16384, 64
buf = torch.zeros(nb * SL, dtype=torch.int64, device=dev)
L.slr_debug_trace(buf.data_ptr())
S.FunctionSoftsplat(x, fl, None, "summation")
torch.cuda.synchronize()
L.slr_debug_trace(None)
t = buf.cpu().numpy().reshape(nb, SL)
t = t[(t[:, 59] > 0) & (t[:, 0] > 0)]
clk = 2.2e3
life = (t[:, 59] - t[:, 0]) / clk
print(f"{which}: {len(t)} workgroups, life us mean {life.mean():.1f} p50 {np.median(life):.1f} p90 {np.percentile(life, 90):.1f} max {life.max():.1f}; sum {life.sum() / 1e3:.1f} ms = {life.sum() / 768:.1f} us on 768 slots")
tw = t[t[:, 57] > 0]
st, en = (tw[:, 56] - tw[:, 56].min()) / 100.0, (tw[:, 57] - tw[:, 56].min()) / 100.0          # wall clock, 100 ticks per us
edges = np.arange(0, min(en.max(), 400.0) + 10, 10.0)
print("  workgroups running at t = 0, 10, 20, ... us: " + " ".join(str(int(((st <= e) & (en > e)).sum())) for e in edges))
pro = (tw[:, 56] - tw[:, 58]) / 100.0
en0 = (tw[:, 58] - tw[:, 56].min()) / 100.0
print("  workgroups resident (entry .. end) at the same times:  " + " ".join(str(int(((en0 <= e) & (en > e)).sum())) for e in edges))
print(f"  kernel entry -> item known: mean {pro.mean():.2f} p50 {np.median(pro):.2f} p90 {np.percentile(pro, 90):.2f} max {pro.max():.2f} us")
print(f"  starts: 768th start at {np.sort(st)[min(767, len(st) - 1)]:.1f} us; last start {st.max():.1f}; last end {en.max():.1f}; lives by wall clock mean {(en - st).mean():.1f}")
def seg(a, b): return (t[:, b] - t[:, a]).mean() / clk
print(f"  item->lists sorted {seg(0, 1):.2f} | rows walk {seg(1, 2):.2f} | entries read + prefetch issue {seg(2, 3):.2f} | footprints + atomics {seg(3, 4):.2f} | "
      f"barrier {seg(4, 5):.2f} | scan + scatter {seg(5, 6):.2f} | lists {seg(6, 7):.2f}")
names = ["stores + wait loads + stage", "barrier", "-", "gather + loads", "-", "barrier"]
for c in range(4):
    b0 = 8 + 6 * c
    prev = 7 if c == 0 else b0 - 1
    parts = [(t[:, b0] - t[:, prev]).mean() / clk] + [(t[:, b0 + k + 1] - t[:, b0 + k]).mean() / clk for k in range(5)]
    print(f"  chunk {c}: " + " | ".join(f"{n} {v:.2f}" for n, v in zip(names, parts)) + f" | total {sum(parts):.2f}")
rest = (t[:, 59] - t[:, 8 + 6 * 7 + 5]).mean() / clk
print(f"  chunks 8..16 {rest:.2f} us ({rest / 9:.2f} each)")
ent, pw = t[:, 60], t[:, 62]
for lo, hi in ((0, 300), (300, 600), (600, 800), (800, 1025)):
    m = (ent >= lo) & (ent < hi)
    if m.any():
        print(f"  entries [{lo},{hi}): {m.sum():6d} workgroups, life mean {life[m].mean():.1f} max {life[m].max():.1f}; width<64: {(pw[m] < 64).sum()}")
