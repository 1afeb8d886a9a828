"""Where a workgroup of the scan front end's tile kernel spends its life (tracing build: make -C slr-sfs_amd/csrc OUT=../lib/var_trace.so DEFS=-DSLR_TRACE).
usage: python tools/dev/trace_scan.py [inc|t30] [h w c]"""
import ctypes, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
os.environ["SLR_SFS_AMD_LIB"] = os.path.join(ROOT, "slr-sfs_amd/lib/var_trace.so")
import slr_sfs_amd as S
from bench import smooth_motion
L = S._lib.lib()
L.slr_debug_trace.argtypes = [ctypes.c_void_p]
L.slr_splat_set_front_end(1)
dev = torch.device("cuda:0")
which = sys.argv[1] if len(sys.argv) > 1 else "inc"
h, w, c = (int(v) for v in sys.argv[2:5]) if len(sys.argv) > 4 else (256, 480, 64)
x, met = torch.randn(1, c, h, w, device=dev), torch.randn(1, 1, h, w, device=dev)
fl = torch.rand(1, 2, h, w, device=dev) * 16 - 8 if which == "inc" else S.euler_integration(torch.from_numpy(smooth_motion(h, w)).to(dev), 30)[0]
S.FunctionSoftsplat(x, fl, met, "softmax")
nb, SL = 16384, 64
buf = torch.zeros(nb * SL, dtype=torch.int64, device=dev)
L.slr_debug_trace(buf.data_ptr())
S.FunctionSoftsplat(x, fl, met, "softmax")
torch.cuda.synchronize()
L.slr_debug_trace(None)
t = buf.cpu().numpy().reshape(nb, SL)
t = t[(t[:, 59] > 0) & (t[:, 0] > 0)]
clk = 2.2e3         # shader clock, cycles per us (approx.)
life = (t[:, 59] - t[:, 0]) / clk
print(f"{which} {c}x{h}x{w}: {len(t)} workgroups streamed, life us mean {life.mean():.1f} p50 {np.median(life):.1f} max {life.max():.1f}; entries mean {t[:, 60].mean():.0f} max {t[:, 60].max()}; candidates mean {t[:, 61].mean():.1f} max {t[:, 61].max()}")
def seg(a, b): return (t[:, b] - t[:, a]).mean() / clk
print(f"  boxes + candidate list {seg(0, 1):.2f} | row walks {seg(1, 2):.2f} | entries read + prefetch issue {seg(2, 3):.2f} | footprints + atomics {seg(3, 4):.2f} | "
      f"barrier {seg(4, 5):.2f} | scan + scatter + lists {seg(5, 6):.2f} | chunks {seg(6, 59):.2f}")
span = (t[:, 59].max() - t[:, 0].min()) / clk
print(f"  first start -> last end {span:.1f} us; starts spread {(t[:, 0].max() - t[:, 0].min()) / clk:.1f} us")
full = buf.cpu().numpy().reshape(nb, SL)
st = full[full[:, 0] > 0]
t00 = st[:, 0].min()
# workgroups that deferred their piece never reach slot 59: their end is unknown; list the longest complete lives and the deferred ones
order = np.argsort(-(t[:, 59] - t[:, 0]))[:8]
for i in order:
    r = t[i]
    print(f"  long: entries {r[60]} candidates {r[61]} | cand {(r[1]-r[0])/clk:.1f} walk {(r[2]-r[1])/clk:.1f} rec {(r[6]-r[2])/clk:.1f} lists {(r[7]-r[6])/clk:.1f} chunks {(r[59]-r[7])/clk:.1f} | life {(r[59]-r[0])/clk:.1f} start {(r[0]-t00)/clk:.1f} longest list {r[63]}")
dfr = st[(st[:, 59] == 0) & (st[:, 2] > 0)]
for r in dfr[np.argsort(-dfr[:, 60])][:6]:
    print(f"  deferred: entries {r[60]} candidates {r[61]} | cand {(r[1]-r[0])/clk:.1f} first walk {(r[2]-r[1])/clk:.1f} second walk + entries out {(r[55]-r[2])/clk:.1f} (0 stamp: a channel group that only returns) life {(r[55]-r[0])/clk:.1f}")
