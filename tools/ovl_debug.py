import sys, torch, torch.nn.functional as F
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
A = torch.randn(2048, 2048, device="cuda")
conv = nets.Conv(64, 128, 3).cuda()
def work_mm(f): return (A @ A).sum()
def work_conv(f): return conv(f)
def work_dec(f): return an.projector(f)
def work_none(f): return None
c1 = nets.Conv(64, 128, 1, bias=False).cuda()
def work_miopen(f): return F.conv2d(f, conv.weight, None, padding=1)
def work_c1(f): return c1(f)
def work_pool(f): return nets.avgpool_down(f)
big = torch.randn(1, 64, 768, 1280, device="cuda")
bigconv = nets.Conv(64, 64, 3).cuda()
def work_bigconv(f): return bigconv(big)
with torch.no_grad():
    for name, work in (("big conv3x3 (full GPU)", work_bigconv), ("decoder", work_dec), ("big conv3x3 again", work_bigconv), ("decoder again", work_dec)):
        bad = 0
        for trial in range(10):
            clip = an.begin_clip(img, m, N)
            feats = []
            for gen_fs in features_ahead_overlap(clip, order):
                feats.append(gen_fs.clone())
                work(gen_fs)
            torch.cuda.synchronize()
            for i, t in enumerate(order):
                if (feats[i] - clip.features(t)).abs().max().item() > 1e-4:
                    bad += 1
        print(f"{name}: wrong feature maps {bad} / {10 * len(order)}", flush=True)
