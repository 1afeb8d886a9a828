"""Aggressor = the 3x3 convolution from a second build of the library (ablation switches), victim = splat on a side stream."""
import sys, ctypes, os, torch
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
buf, wscale = bigconv._split_weights()
outbuf = torch.empty(1, 64, 768, 1280, device="cuda")
vp, f, i = ctypes.c_void_p, ctypes.c_float, ctypes.c_int
for libname in sys.argv[1:]:
    Lb = ctypes.CDLL(os.path.abspath(libname))
    Lb.slr_conv3x3_forward.argtypes = [vp, vp, vp, vp, vp, i, i, i, i, i, f, vp, vp, i, vp]
    Lb.slr_conv3x3_forward.restype = i
    def work(_):
        rc = Lb.slr_conv3x3_forward(big.data_ptr(), buf.data_ptr(), None, None, outbuf.data_ptr(), 1, 64, 64, 768, 1280,
                                    wscale, None, None, 0, torch.cuda.current_stream().cuda_stream)
        assert rc == 0
    bad = 0
    with torch.no_grad():
        for trial in range(30):
            clip = an.begin_clip(img, m, N)
            feats = []
            for gen_fs in features_ahead_overlap(clip, order):
                feats.append(gen_fs.clone())
                work(gen_fs)
            torch.cuda.synchronize()
            for k, t in enumerate(order):
                if (feats[k] - clip.features(t)).abs().max().item() > 1e-4:
                    bad += 1
    print(f"aggressor {libname}: wrong feature maps {bad} / {30 * len(order)}", flush=True)
