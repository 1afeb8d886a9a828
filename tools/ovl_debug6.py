"""Victim-side experiments for the open concurrency issue: run with SLR_SFS_AMD_LIB=<variant build of the library>
(make OUT=... DEFS=-DSLR_DBG=n).  Counts frames whose side-stream features differ from the sequential ones, and
values carrying the 12345 marker of the SLR_DBG=4 build (staged LDS value != global memory)."""
import sys, torch
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import slr_sfs_amd as S
sys.path.insert(0, '/root/repo/tools')
from ovl_common import features_ahead_overlap
from slr_sfs_amd import nets, _lib
from test_gpu_parity import smooth_motion, dev
H, W, N = 40, 72, 7
torch.manual_seed(1)
an = S.pipeline.BaselineAnimator().cuda().eval()
img = torch.rand(1, 3, H, W, device="cuda") * 2 - 1
m = dev(smooth_motion(H, W, 5, amp=2.0))
order = [0, 2, 3, 6, 1, 4, 5]
big = torch.randn(1, 64, 768, 1280, device="cuda")
bigconv = nets.Conv(64, 64, 3).cuda()
frames = wrong = marked = seq_marked = 0
with torch.no_grad():
    for trial in range(int(sys.argv[1]) if len(sys.argv) > 1 else 40):
        clip = an.begin_clip(img, m, N)
        feats = []
        for gen_fs in features_ahead_overlap(clip, order):
            feats.append(gen_fs.clone())
            bigconv(big)
        torch.cuda.synchronize()
        for i, t in enumerate(order):
            ref = clip.features(t)
            frames += 1
            wrong += int((feats[i] - ref).abs().max().item() > 1e-4)
            marked += int((feats[i] == 12345.0).sum())
            seq_marked += int((ref == 12345.0).sum())
print(f"{_lib.LIB_PATH}: frames {frames}, wrong {wrong}, marker values concurrent {marked} sequential {seq_marked}")
