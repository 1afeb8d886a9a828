#!/usr/bin/env python
"""Animate every scene of a directory -- counterpart of the reference's
    python test_animating/CLAW/test_all_CLAW_scenes.py IMAGE_DIR FLOW_DIR SAVE_DIR CKPT NAME W N SPEED PYFILE SCENE ALIGNFILE START END
(:1-96: ``<scene>_input.jpg`` + ``<scene>.flo`` pairs, one run of the per-scene script each) with the same leading
arguments.  One process: the reference's flow.  Under torchrun --nproc-per-node G: config C5 of BASELINE.json -- the
frames of every clip sharded over the G GPUs of the node, one all-gather per clip, rank 0 writes
SAVE_DIR/<scene>/<scene>/PredImg/%06d.png (the reference's nesting: save_dir/name, then /name inside the per-scene script).
The model is built and the checkpoint read once, not once per scene."""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from animate import init_ranks  # noqa: E402
from slr_sfs_amd import runner  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("image_dir"), ap.add_argument("flow_dir"), ap.add_argument("save_dir"), ap.add_argument("ckpt")
    ap.add_argument("name", nargs="?", default="Demo")          # (the reference overwrites it with the scene name, :80)
    ap.add_argument("W", nargs="?", type=int, default=256), ap.add_argument("N", nargs="?", type=int, default=60)
    ap.add_argument("speed", nargs="?", type=float, default=0.25)
    ap.add_argument("align", nargs="?", default="None"), ap.add_argument("start", nargs="?", type=int, default=-1)
    ap.add_argument("end", nargs="?", type=int, default=-1)
    ap.add_argument("--H", type=int, default=None), ap.add_argument("--v1", action="store_true")
    ap.add_argument("--no-video", action="store_true")
    ap.add_argument("--shard", default="frames", choices=["frames", "scenes"],
                    help="under torchrun: frames of every clip over the ranks + one all-gather per clip (default; config C5), or "
                         "whole scenes per rank, no collective (the reference's own parallelism: one process per scene, test_sbatch_2.sh)")
    ap.add_argument("--half-size", action="store_true", help="write frames at half the raw size (test_baseline_4eval.py / test_v1_4eval.py)")
    a = ap.parse_args()
    rank, world, dev = init_ranks()
    model = runner.load_model(a.ckpt, a.v1, dev)
    scenes = runner.list_scenes(a.image_dir, a.flow_dir, a.align, a.start, a.end)
    busy, t0 = 0.0, time.perf_counter()
    by_scene = a.shard == "scenes" and world > 1
    for i, (scene, img, flo) in enumerate(scenes):
        if by_scene and i % world != rank:
            continue                                             # another rank's scene
        r, w = (0, 1) if by_scene else (rank, world)             # a scene of one's own is rendered and written alone
        dt, out = runner.animate_scene(model, img, flo, os.path.join(a.save_dir, scene), scene, a.H or a.W, a.W, a.N, a.speed,
                                       a.align, r, w, video=not a.no_video, half_size=a.half_size)
        busy += dt
        if r == 0:
            print(f"{scene}: {a.N} frames in {dt:.2f} s ({a.N / dt:.1f} frames/s) -> {out}", flush=True)
    if world > 1:
        torch.distributed.barrier()
    if rank == 0:
        n = len(scenes) * a.N
        wall = time.perf_counter() - t0
        print(f"{len(scenes)} scenes, {n} frames on {world} GPU(s) ({a.shard} sharded): "
              + (f"{n / max(busy, 1e-9):.1f} frames/s rendering, " if not by_scene else "")
              + f"{wall:.1f} s with loading and PNG writing ({n / wall:.1f} frames/s)")
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
