import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
import slr_sfs_amd as S
from slr_sfs_amd import pipeline
from bench import smooth_motion, H, W, NFRAMES
dev = torch.device("cuda:0")
torch.manual_seed(0)
model = pipeline.BaselineAnimator().to(dev).eval()
rng = np.random.default_rng(0)
image = torch.from_numpy(rng.uniform(-1, 1, (1, 3, H, W)).astype(np.float32)).to(dev)
motion = torch.from_numpy(smooth_motion(H, W)).to(dev)
with torch.no_grad():
    model.synthesize(image, motion, NFRAMES)
    torch.cuda.synchronize()
    from torch.profiler import profile, ProfilerActivity
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True, record_shapes=True) as prof:
        model.synthesize(image, motion, NFRAMES)
        torch.cuda.synchronize()
evs = [e for e in prof.events() if "emcpy" in e.name or "copy_" in e.name.lower() or "copyBuffer" in e.name]
from collections import Counter
c = Counter()
for e in evs:
    st = [s for s in (e.stack or []) if "slr-sfs_amd" in s or "bench" in s]
    c[(e.name[:40], str(getattr(e, "input_shapes", ""))[:80], st[0][-80:] if st else "")] += 1
for k, v in c.most_common(25):
    print(v, k)
print(prof.key_averages().table(sort_by="cuda_time_total", row_limit=12, max_name_column_width=60))
