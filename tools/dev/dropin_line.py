"""bench.py's roofline_dropin alone (graph-replayed calls), one line: python tools/dev/dropin_line.py"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
dev = torch.device("cuda:0")
motion = torch.from_numpy(bench.smooth_motion(bench.H, bench.W)).to(dev)
d = bench.dropin_roofline(dev, motion)
print(os.path.basename(os.environ.get("SLR_SFS_AMD_LIB", "default")), " | ".join(f"{k} {v['call_us']:.1f}" for k, v in d["flows"].items()),
      "| c2", d["c2"]["call_us"], "c2b", d["c2_batched"]["per_sample_us"], "|", " ".join(f"{k} {v['call_us']:.1f}" for k, v in d["small_grids"].items()))
