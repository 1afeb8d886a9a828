"""Both all-frames Euler passes of a clip (768x1280, N = 60) -- us per direction.  usage: python tools/dev/euler_bench.py"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import slr_sfs_amd as S
from bench import smooth_motion, H, W, NFRAMES
from kbench import timeit
m = torch.from_numpy(smooth_motion(H, W)).cuda()
t = timeit(lambda: S.euler_integration_all(m, NFRAMES, want_visible=False), 20)
print(os.path.basename(os.environ.get("SLR_SFS_AMD_LIB", "default")), "euler_integration_all(+M, 60) us", t)
