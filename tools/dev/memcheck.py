"""Device memory over repeated clips (allocated / reserved after each): no growth, with and without the two-stream option."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))); sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tools")); sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
import sys, torch
pass
import slr_sfs_amd as S
from test_gpu_parity import smooth_motion, dev
H, W, N = 768, 1280, 60
torch.manual_seed(0)
for name, an in (("baseline", S.pipeline.BaselineAnimator().cuda().eval()), ("slr-v1", S.pipeline.SLRv1Animator().cuda().eval())):
    img = torch.rand(1, 3, H, W, device="cuda") * 2 - 1
    m = dev(smooth_motion(H, W, 5, amp=1.5))
    for overlap in (False, True):
        rows = []
        for rep in range(6):
            out = an.synthesize(img, m, N, overlap=overlap)
            torch.cuda.synchronize()
            del out
            rows.append((torch.cuda.memory_allocated() / 2**30, torch.cuda.memory_reserved() / 2**30))
        print(name, "overlap" if overlap else "one stream", " ".join(f"{a:.2f}/{r:.2f}" for a, r in rows), "GiB allocated/reserved", flush=True)
