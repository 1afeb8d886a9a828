"""Where a workgroup of the Winograd fp32 convolution spends its life (tracing build: make -C slr-sfs_amd/csrc OUT=../lib/var_trace.so DEFS=-DSLR_TRACE).
usage: python tools/dev/trace_wino.py [cin cout h w]"""
import ctypes, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
os.environ.setdefault("SLR_SFS_AMD_LIB", os.path.join(ROOT, "slr-sfs_amd/lib/var_trace.so"))
import slr_sfs_amd as S
from slr_sfs_amd import nets
L = S._lib.lib()
L.slr_debug_trace.argtypes = [ctypes.c_void_p]
cin, cout, h, w = (int(v) for v in sys.argv[1:5]) if len(sys.argv) >= 5 else (128, 128, 768, 1280)
conv = nets.Conv(cin, cout, 3).cuda()
x = torch.randn(1, cin, h, w, device="cuda")
nb = 1 << 16
LAY = 0 if '--nchw' in sys.argv else nets.IN_B8 | nets.OUT_B8        # the networks' activations are channel-blocked
buf = torch.zeros(nb * 16, dtype=torch.int64, device="cuda")
with torch.no_grad(), nets.fp32_kernels(winograd=True):
    for _ in range(3):
        y = conv(x, layout=LAY)
    torch.cuda.synchronize()
    L.slr_debug_trace(buf.data_ptr())
    y = conv(x, layout=LAY)
    torch.cuda.synchronize()
    L.slr_debug_trace(None)
t = buf.cpu().numpy().reshape(nb, 16)
t = t[(t[:, 0] > 0) & (t[:, 11] > 0)]
clk = 2.2e3                                     # clock64(): shader clock, cycles per us (approx.)
t0 = t[:, 0].min()
life = (t[:, 11] - t[:, 0]) / clk
print(f"{len(t)} workgroups; kernel span {(t[:, 11].max() - t0) / clk:.1f} us; life mean {life.mean():.2f} p10 {np.percentile(life, 10):.2f} p50 {np.median(life):.2f} p90 {np.percentile(life, 90):.2f} us; "
      f"sum / 512 slots {life.sum() / 512:.1f} us")
def seg(a, b): return (t[:, b] - t[:, a]) / clk
nch = (cin + 15) // 16
last = 1 + min(nch, 8)
print(f"  prologue {seg(0, 1).mean():.2f} | chunks " + " ".join(f"{seg(1 + c, 2 + c).mean():.2f}" for c in range(min(nch, 8)))
      + f" | exchange {seg(last, 10).mean():.2f} | epilogue: setup {0.0:.2f}, rows 0-3 {seg(10, 14).mean():.2f}, stores + rows 4-7 {seg(14, 15).mean():.2f}, stores {seg(15, 11).mean():.2f}")
# co-residency: same (xcc, se, cu) -> sort by start, gap between the end of a workgroup and the start of the next in the same slot
hw = t[:, 12]
cu = ((hw >> 32) << 16) | (hw & 0xffffffff & ~0xf & ~(0x3 << 4))      # drop wave / simd id bits (wave_id[3:0], simd_id[5:4])
order = np.lexsort((t[:, 0], cu))
ts, cs = t[order], cu[order]
gaps = []
for k in np.unique(cs):
    m = ts[cs == k]
    # two slots per CU: greedy assignment
    ends = []
    for row in m:
        best = None
        for i, e in enumerate(ends):
            if e <= row[0] and (best is None or e > ends[best]): best = i
        if best is None: ends.append(row[11])
        else:
            gaps.append((row[0] - ends[best]) / clk); ends[best] = row[11]
gaps = np.array(gaps)
print(f"  {len(np.unique(cs))} CUs seen; gap between a workgroup's last stamp and the next one's first in the same slot: mean {gaps.mean():.2f} p50 {np.median(gaps):.2f} p90 {np.percentile(gaps, 90):.2f} us")
# overlap: for a sample CU print the timeline
k = np.unique(cs)[5]
m = ts[cs == k][:10] if '-v' in sys.argv else []
for row in m:
    print("   ", " ".join(f"{(v - t0) / clk:7.2f}" for v in row[:12]))
