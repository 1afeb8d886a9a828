#!/usr/bin/env python
"""Animate one still image on MI355X -- counterpart of the reference's
    python test_animating/test_baseline_4eval_rawsize.py IMG FLOW OUTDIR CKPT NAME W N SPEED ALIGN
(test_animating/CLAW/test_all_CLAW_scenes.py:86-96) with the same positional arguments; writes
OUTDIR/NAME/PredImg/%06d.png (--v1: also FluidImg/, CompositeFluidAlpha/, BGImg.png).  Without a checkpoint (CKPT = None)
the networks are random-initialised (plumbing / timing only).  --v1 runs the 2-layer SLR model (test_v1_4eval_rawsize.py).
Under torchrun the frames of the clip are rendered by all ranks (slr_sfs_amd/runner.py)."""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from slr_sfs_amd import runner  # noqa: E402


def init_ranks():
    """(rank, world, device): one process per GPU under torchrun (RCCL), a single process otherwise."""
    import torch.distributed as dist
    rank, world, local = (int(os.environ.get(k, d)) for k, d in (("RANK", 0), ("WORLD_SIZE", 1), ("LOCAL_RANK", 0)))
    one_gpu = os.environ.get("SLR_ONE_GPU_GLOO") == "1"        # development: all ranks on cuda:0, collectives over gloo
    dev = torch.device("cuda", 0 if one_gpu else local)
    torch.cuda.set_device(dev)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if one_gpu:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    return rank, world, dev


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("image"), ap.add_argument("flow"), ap.add_argument("outdir"), ap.add_argument("ckpt")
    ap.add_argument("name"), ap.add_argument("W", type=int), ap.add_argument("N", type=int)
    ap.add_argument("speed", type=float), ap.add_argument("align", nargs="?", default="None")
    ap.add_argument("--H", type=int, default=None, help="working height (default: W, square like the reference)")
    ap.add_argument("--v1", action="store_true")
    ap.add_argument("--half-size", action="store_true", help="write frames at half the raw size (test_baseline_4eval.py / test_v1_4eval.py)")
    a = ap.parse_args()
    rank, world, dev = init_ranks()
    model = runner.load_model(a.ckpt, a.v1, dev)
    dt, out = runner.animate_scene(model, a.image, a.flow, a.outdir, a.name, a.H or a.W, a.W, a.N, a.speed, a.align, rank, world,
                                   half_size=a.half_size)
    if rank == 0:
        print(f"{a.N} frames at {a.H or a.W}x{a.W} on {world} GPU(s) in {dt:.2f} s ({a.N / dt:.1f} frames/s) -> {out}")
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
