"""Frames/s of the C3 clip with frame i+1's splat on a side stream under frame i's decoder vs everything on one stream."""
import sys, time, torch
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tools'); sys.path.insert(0, '/root/repo/tests')
import slr_sfs_amd as S
from ovl_common import features_ahead_overlap
from test_gpu_parity import smooth_motion, dev
H, W, N = 768, 1280, 60
torch.manual_seed(0)
an = S.pipeline.BaselineAnimator().cuda().eval()
img = torch.rand(1, 3, H, W, device="cuda") * 2 - 1
m = dev(smooth_motion(H, W, 5, amp=1.5))
out = torch.empty(N, 3, H, W, device="cuda")
def run(overlap):
    clip = an.begin_clip(img, m, N)
    gen = features_ahead_overlap(clip, range(N)) if overlap else (clip.features(t) for t in range(N))
    for i, fs in enumerate(gen):
        out[i] = torch.tanh(an.projector(fs))[0]
with torch.no_grad():
    ref = None
    for mode in (False, True, False, True):
        run(mode); torch.cuda.synchronize()
        t0 = time.perf_counter(); run(mode); torch.cuda.synchronize(); dt = time.perf_counter() - t0
        if ref is None: ref = out.clone()
        print(f"overlap={mode}: {N / dt:.1f} frames/s, max |diff| vs first run {(out - ref).abs().max().item():.2e}", flush=True)
