"""Characterise the wrong values of the concurrency issue (see tools/ovl_debug2.py): value, reference, location."""
import sys, torch
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import slr_sfs_amd as S
sys.path.insert(0, '/root/repo/tools')
from ovl_common import features_ahead_overlap
from slr_sfs_amd import nets
from test_gpu_parity import smooth_motion, dev
H, W, N = 40, 72, 7
torch.manual_seed(1)
an = S.pipeline.BaselineAnimator().cuda().eval()
img = torch.rand(1, 3, H, W, device="cuda") * 2 - 1
m = dev(smooth_motion(H, W, 5, amp=2.0))
order = [0, 2, 3, 6, 1, 4, 5]
big = torch.randn(1, 64, 768, 1280, device="cuda")
bigconv = nets.Conv(64, 64, 3).cuda()
shown = 0
with torch.no_grad():
    for trial in range(40):
        clip = an.begin_clip(img, m, N)
        feats = []
        for gen_fs in features_ahead_overlap(clip, order):
            feats.append(gen_fs.clone())
            bigconv(big)
        torch.cuda.synchronize()
        for i, t in enumerate(order):
            ref = clip.features(t)
            d = (feats[i] - ref).abs()
            if d.max().item() > 1e-4 and shown < 5:
                shown += 1
                bad = (d > 1e-4)[0]
                idx = bad.nonzero()
                print(f"trial {trial} frame {t}: {idx.shape[0]} wrong", flush=True)
                pix = {}
                for c, y, x in idx.tolist():
                    pix.setdefault((y, x), []).append(c)
                for (y, x), cs in list(pix.items())[:6]:
                    g = feats[i][0, :, y, x]; r = ref[0, :, y, x]
                    print(f"  pixel ({y},{x}) tile ({y // 16},{x // 16}) local ({y % 16},{x % 16}): planes {cs[:12]} n={len(cs)}")
                    for c in cs[:4]:
                        print(f"     c={c}: got {g[c].item():+.6f} ref {r[c].item():+.6f} diff {(g[c]-r[c]).item():+.6f}  "
                              f"neighbours ref c-1 {r[c-1].item():+.6f} c+1 {r[min(c+1, r.numel()-1)].item():+.6f}")
print("done")
