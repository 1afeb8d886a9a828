"""hipGraph capture (torch.cuda.CUDAGraph) of one frame of the pipeline vs eager launches: the frame is GPU-bound even at
256x480, so replay buys nothing (1.48 vs 1.47 ms; 6.61 vs 6.58 ms at 768x1280) -- but it shows that a frame is capturable:
every launch on the caller's stream, no allocation or synchronisation inside."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))); sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tools")); sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
import sys, time, torch
pass
import slr_sfs_amd as S
from test_gpu_parity import smooth_motion, dev
for (H, W) in ((256, 480), (768, 1280)):
    N = 60
    torch.manual_seed(0)
    an = S.pipeline.BaselineAnimator().cuda().eval()
    img = torch.rand(1, 3, H, W, device="cuda") * 2 - 1
    m = dev(smooth_motion(H, W, 5, amp=1.5))
    with torch.no_grad():
        clip = an.begin_clip(img, m, N)
        t = 30
        ref = an.frame(clip, t).clone()
        torch.cuda.synchronize()
        # eager timing
        t0 = time.perf_counter()
        for _ in range(20): out = an.frame(clip, t)
        torch.cuda.synchronize(); e = (time.perf_counter() - t0) / 20
        # capture
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            for _ in range(2): an.frame(clip, t)
        torch.cuda.current_stream().wait_stream(s)
        g = torch.cuda.CUDAGraph()
        try:
            with torch.cuda.graph(g):
                gout = an.frame(clip, t)
            g.replay(); torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(20): g.replay()
            torch.cuda.synchronize(); gt = (time.perf_counter() - t0) / 20
            print(f"{H}x{W}: eager {e*1e3:.3f} ms/frame, graph replay {gt*1e3:.3f} ms/frame, max diff {(gout-ref).abs().max().item():.2e}", flush=True)
        except Exception as ex:
            print(f"{H}x{W}: capture failed: {type(ex).__name__}: {str(ex)[:300]}", flush=True)
