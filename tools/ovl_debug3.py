import sys, torch
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import slr_sfs_amd as S
from slr_sfs_amd import nets, synthesis
from test_gpu_parity import smooth_motion, dev
H, W, N = 40, 72, 7
torch.manual_seed(1)
an = S.pipeline.BaselineAnimator().cuda().eval()
img = torch.rand(1, 3, H, W, device="cuda") * 2 - 1
m = dev(smooth_motion(H, W, 5, amp=2.0))
big = torch.randn(1, 64, 768, 1280, device="cuda")
bigconv = nets.Conv(64, 64, 3).cuda()
side = torch.cuda.Stream()
main = torch.cuda.current_stream()
with torch.no_grad():
    clip = an.begin_clip(img, m, N)
    torch.cuda.synchronize()
    for mode in ("bins concurrent, splat alone", "bins alone, splat concurrent"):
        bad = 0
        for trial in range(60):
            t = 1 + trial % 6
            ref = clip.features(t)
            torch.cuda.synchronize()
            disp_f = clip.disp_f[t:t + 1]; disp_p = clip.disp_p[N - t:N - t + 1]
            if mode.startswith("bins concurrent"):
                bigconv(big)
                with torch.cuda.stream(side):
                    ws_f, ws_p = synthesis.bin_flow_pair(disp_f, disp_p, clip.C)
                torch.cuda.synchronize()
                with torch.cuda.stream(side):
                    out = synthesis.synth_group(clip.fs, clip.Z, disp_f, disp_p, clip.alpha(t), ws_f, ws_p, wmax=clip.zmax)
                torch.cuda.synchronize()
            else:
                with torch.cuda.stream(side):
                    ws_f, ws_p = synthesis.bin_flow_pair(disp_f, disp_p, clip.C)
                torch.cuda.synchronize()
                bigconv(big)
                with torch.cuda.stream(side):
                    out = synthesis.synth_group(clip.fs, clip.Z, disp_f, disp_p, clip.alpha(t), ws_f, ws_p, wmax=clip.zmax)
                torch.cuda.synchronize()
            if (out - ref).abs().max().item() > 1e-4:
                bad += 1
        print(f"{mode}: wrong {bad} / 60", flush=True)
