"""Where the workgroups of the scan front end's SINK launch spend their lives (tracing build: make -C slr-sfs_amd/csrc OUT=../lib/var_trace.so DEFS=-DSLR_TRACE).
usage: python tools/dev/trace_sink.py [t30|t59] [n h w c]"""
import ctypes, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
os.environ["SLR_SFS_AMD_LIB"] = os.path.join(ROOT, "slr-sfs_amd/lib/var_trace.so")
import slr_sfs_amd as S
from bench import smooth_motion
L = S._lib.lib()
L.slr_debug_trace.argtypes = [ctypes.c_void_p]
dev = torch.device("cuda:0")
which = sys.argv[1] if len(sys.argv) > 1 else "t30"
n, h, w, c = (int(v) for v in sys.argv[2:6]) if len(sys.argv) > 5 else (1, 256, 480, 64)
x, met = torch.randn(n, c, h, w, device=dev), torch.randn(n, 1, h, w, device=dev)
mo = torch.from_numpy(np.concatenate([smooth_motion(h, w, seed=i) for i in range(n)], 0)).to(dev)
fl = S.EulerIntegration()(mo, torch.full((n,), int(which[1:]), device=dev)).contiguous()
S.FunctionSoftsplat(x, fl, met, "softmax")
nb = 16384
buf = torch.zeros(nb * 64 + 8192 * 16, dtype=torch.int64, device=dev)
L.slr_debug_trace(buf.data_ptr())
S.FunctionSoftsplat(x, fl, met, "softmax")
torch.cuda.synchronize()
L.slr_debug_trace(None)
traw = buf.cpu().numpy()[nb * 64:].reshape(-1, 16)
t = traw[traw[:, 0] > 0]
us = 100.0          # wall clock: 100 MHz
t0 = t[:, 0].min()
fin = t[:, 4] > 0
print(f"{which} {n}x{c}x{h}x{w}: {len(t)} sink workgroups with planes, {int(fin.sum())} reached the end of a piece; span first start -> last end {(t[fin, 4].max() - t0) / us:.1f} us; "
      f"starts spread {(t[:, 0].max() - t0) / us:.1f} us")
w_ = t[fin]
seg = lambda a, b, m=slice(None): ((w_[m, b] - w_[m, a]) / us)
print(f"  candidates {seg(0, 1).mean():.1f} | tasks {seg(1, 2).mean():.1f} (max {seg(1, 2).max():.1f}; tasks per workgroup mean {w_[:, 9].mean():.2f} max {w_[:, 9].max()}; candidates mean {w_[:, 8].mean():.1f} max {w_[:, 8].max()}) | "
      f"arrive {seg(2, 3).mean():.1f} | after arrival (finalise when last) {seg(3, 4).mean():.1f}")
last = w_[:, 10] == 1
print(f"  finalisers: {int(last.sum())}, finalise us mean {seg(3, 4, last).mean():.1f} max {seg(3, 4, last).max():.1f}; task us per task {(seg(1, 2) / np.maximum(w_[:, 9], 1))[w_[:, 9] > 0].mean():.1f}")
print(f"  start offsets us p50 {np.median((w_[:, 0] - t0) / us):.1f} p90 {np.percentile((w_[:, 0] - t0) / us, 90):.1f} max {((w_[:, 0] - t0) / us).max():.1f}; life mean {seg(0, 4).mean():.1f} max {seg(0, 4).max():.1f}")
end = np.where(t[:, 4] > 0, t[:, 4], t[:, 0] + 30)       # (workgroups without a piece: ~0.3 us)
act = t[:, 4] > 0
prof = []
for tt in range(0, int((end.max() - t0) / us) + 1, 10):
    x = t0 + tt * us
    prof.append((tt, int(((t[:, 0] <= x) & (end > x) & act).sum())))
print("  active workgroups alive at t (us):", " ".join(f"{a}:{b}" for a, b in prof))
order = np.argsort(-(w_[:, 4] - w_[:, 0]))[:6]
for i in order:
    r = w_[i]
    print(f"  long: piece {r[11]} entries/candidates {r[8]} tasks {r[9]} last {r[10]} | setup {(r[1]-r[0])/us:.1f} tasks {(r[2]-r[1])/us:.1f} (last task: entries in {(r[5]-r[1])/us:.1f}, records {(r[6]-r[5])/us:.1f}, planes {(r[7]-r[6])/us:.1f}) arrive {(r[3]-r[2])/us:.1f} finalise {(r[4]-r[3])/us:.1f} start {(r[0]-t0)/us:.1f}")
tt = buf.cpu().numpy()[:nb * 64].reshape(nb, 64)
gx = 32 if n * ((h + 7) // 8) * ((w + 63) // 64) >= 32 else n * ((h + 7) // 8) * ((w + 63) // 64)
for y in range(2):
    r = tt[y * gx + 0]
    names = {3: "entries read", 4: "footprints+atomics", 5: "barrier", 6: "scan+scatter", 7: "pixel lists+norm", 8: "chunk0 staged", 9: "c0 barrier", 11: "c0 gathered", 13: "c0 barrier2", 14: "chunk1 staged", 59: "end"}
    ks = [k for k in sorted(names) if r[k] > 0]
    print(f"  tile-trace of sink block x=0 y={y} (shader clock, 2.2 cycles/ns; the LAST task that ran there):", " | ".join(f"{names[b]} +{(r[b] - r[a]) / 2200:.1f}" for a, b in zip(ks, ks[1:])), f"| list len note {r[63]}")
# the tile-kernel stamps (shader clock) of the (x, y) block of the longest sink workgroup: grid = 33 (or fewer) pieces x groups x 16 slots
idx = int(np.argmax(np.where(traw[:, 4] > 0, traw[:, 4] - traw[:, 0], 0)))
gxy = None
for gy in range(1, 9):
    if (len(traw) and idx // (gx * gy) < 16): gxy = gy
xy = idx % (gx * gxy) if gxy else 0
r = tt[xy]
ks = [k for k in sorted(names) if r[k] > 0]
print(f"  tile-trace of the longest sink workgroup's block (index {idx}, x+gx*y = {xy}):", " | ".join(f"{names[b]} +{(r[b] - r[a]) / 2200:.1f}" for a, b in zip(ks, ks[1:])))
