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
    for trial in range(30):
        clip = an.begin_clip(img, m, N)
        feats = []
        for gen_fs in features_ahead_overlap(clip, order):
            feats.append(gen_fs.clone())
            bigconv(big)
        torch.cuda.synchronize()
        for i, t in enumerate(order):
            ref = clip.features(t)
            d = (feats[i] - ref).abs()
            if d.max().item() > 1e-4 and shown < 6:
                shown += 1
                bad = d > 1e-4
                planes = bad.flatten(2).any(2)[0].nonzero().flatten().tolist()
                ys, xs = bad.any(1)[0].nonzero(as_tuple=True)
                print(f"trial {trial} frame {t}: {int(bad.sum())} wrong values, planes {planes[:8]}{'...' if len(planes) > 8 else ''} ({len(planes)}), "
                      f"rows {ys.min().item()}..{ys.max().item()} cols {xs.min().item()}..{xs.max().item()}, max diff {d.max().item():.3g}, "
                      f"wrong==0: {int((feats[i][bad] == 0).sum())}, ref==0: {int((ref[bad] == 0).sum())}", flush=True)
print("done")
