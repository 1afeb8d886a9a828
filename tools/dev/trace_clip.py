"""Where a workgroup of the fused clip kernel spends its life: per-phase shader-clock stamps of every workgroup of one 8-frame launch
(tracing build: make -C slr-sfs_amd/csrc OUT=../lib/var_trace.so DEFS=-DSLR_TRACE).  usage: python tools/dev/trace_clip.py [first frame]"""
import ctypes, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
os.environ["SLR_SFS_AMD_LIB"] = os.path.join(ROOT, "slr-sfs_amd/lib/var_trace.so")
import slr_sfs_amd as S
from bench import smooth_motion, H, W, NFRAMES
L = S._lib.lib()
L.slr_debug_trace.argtypes = [ctypes.c_void_p]
dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(0)
fs = torch.randn(1, 64, H, W, device=dev, generator=g)
Z = torch.randn(1, 1, H, W, device=dev, generator=g)
cs = S.synthesis.ClipSynthesizer(fs, Z, torch.from_numpy(smooth_motion(H, W)).to(dev), NFRAMES)
t0 = int(sys.argv[1]) if len(sys.argv) > 1 else 24
ts = list(range(t0, t0 + 8))
out = torch.empty(8, 64, H, W, device=dev)
cs.features_batch(ts, out)
nb, SL = 32768, 64
buf = torch.zeros(nb * SL, dtype=torch.int64, device=dev)
L.slr_debug_trace(buf.data_ptr())
cs.features_batch(ts, out)
torch.cuda.synchronize()
L.slr_debug_trace(None)
t = buf.cpu().numpy().reshape(nb, SL)
t = t[(t[:, 59] > 0) & (t[:, 0] > 0)]
clk = 2.2e3                                    # shader clock, cycles per us (approx.)
life = (t[:, 59] - t[:, 0]) / clk
print(f"frames {ts[0]}..{ts[-1]}: {len(t)} workgroups, life us mean {life.mean():.1f} p50 {np.median(life):.1f} p90 {np.percentile(life, 90):.1f} max {life.max():.1f}; "
      f"sum {life.sum() / 1e3:.1f} ms = {life.sum() / 512 / 8:.1f} us per frame on 512 slots")
def seg(a, b): return (t[:, b] - t[:, a]).mean() / clk
print(f"  item->lists sorted {seg(0, 1):.2f} | rows walk {seg(1, 2):.2f} | entries read + prefetch issue {seg(2, 3):.2f} | Z + footprints + atomics {seg(3, 4):.2f} | "
      f"barrier {seg(4, 5):.2f} | scan + scatter {seg(5, 6):.2f} | lists + normaliser {seg(6, 7):.2f}")
names = ["stores + wait loads + stage", "barrier", "-", "gather + loads", "-", "barrier"]
for c in range(8):
    b0 = 8 + 6 * c
    prev = 7 if c == 0 else b0 - 1
    parts = [(t[:, b0] - t[:, prev]).mean() / clk] + [(t[:, b0 + k + 1] - t[:, b0 + k]).mean() / clk for k in range(5)]
    print(f"  chunk {c}: " + " | ".join(f"{n} {v:.2f}" for n, v in zip(names, parts)) + f" | total {sum(parts):.2f}")
rest = (t[:, 59] - t[:, 8 + 6 * 7 + 5]).mean() / clk
print(f"  chunks 8..15 {rest:.2f} us ({rest / 8:.2f} each)")
ent, pw, rl = t[:, 60], t[:, 62], t[:, 63]
for lo, hi in ((0, 600), (600, 1000), (1000, 1300), (1300, 1537)):
    m = (ent >= lo) & (ent < hi)
    if m.any():
        print(f"  entries [{lo},{hi}): {m.sum():6d} workgroups, life mean {life[m].mean():.1f} max {life[m].max():.1f}; width<64: {(pw[m] < 64).sum()}")
