#!/usr/bin/env python
"""Animate one still image on MI355X -- counterpart of the reference's
    python test_animating/test_baseline_4eval_rawsize.py IMG FLOW OUTDIR CKPT NAME W N SPEED ALIGN
(test_animating/CLAW/test_all_CLAW_scenes.py:86-96) with the same positional arguments; writes
OUTDIR/NAME/PredImg/%06d.png (--v1: also FluidImg/, CompositeFluidAlpha/, BGImg.png).  Without a checkpoint (CKPT = None) the networks are random-initialised
(plumbing / timing only).  --v1 runs the 2-layer SLR model (test_v1_4eval_rawsize.py)."""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import slr_sfs_amd as S  # noqa: E402
from slr_sfs_amd import io, nets, pipeline  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("image"), ap.add_argument("flow"), ap.add_argument("outdir"), ap.add_argument("ckpt")
    ap.add_argument("name"), ap.add_argument("W", type=int), ap.add_argument("N", type=int)
    ap.add_argument("speed", type=float), ap.add_argument("align", nargs="?", default="None")
    ap.add_argument("--H", type=int, default=None, help="working height (default: W, square like the reference)")
    ap.add_argument("--v1", action="store_true")
    a = ap.parse_args()
    H = a.H or a.W
    dev = torch.device("cuda:0")
    model = (pipeline.SLRv1Animator() if a.v1 else pipeline.BaselineAnimator()).to(dev).eval()
    if a.ckpt not in ("None", "none", ""):
        sd = torch.load(a.ckpt, map_location="cpu", weights_only=False)["state_dict"]
        pre = "model.module."
        nets.load_reference_state_dict(model.encoder, sd, pre + "encoder.")
        nets.load_reference_state_dict(model.projector, sd, pre + "projector.")
        if a.v1:
            nets.load_reference_state_dict(model.net_bg, sd, pre + "net_bg.")
            nets.load_reference_state_dict(model.net_alpha_encoder, sd, pre + "net_alpha_encoder.")
            nets.load_reference_state_dict(model.net_alpha_decoder, sd, pre + "net_alpha_decoder.")
    image, (raw_w, raw_h) = io.load_image(a.image, H, a.W)
    motion = pipeline.prepare_motion(io.load_motion(a.flow), H, a.W, a.speed, io.speed_align(a.align, a.name), a.N)
    t0 = time.perf_counter()
    if a.v1:
        outs = model.synthesize(image.to(dev), motion.to(dev), a.N, keys=pipeline.SLRv1Animator.KEYS)
        frames = outs["PredImg"]
    else:
        frames = model.synthesize(image.to(dev), motion.to(dev), a.N)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    scene = os.path.join(a.outdir, a.name)
    out = io.save_frames(io.frames_to_uint8(frames, (raw_h, raw_w)), scene)
    if a.v1:        # test_v1_4eval_rawsize.py:240-284: FluidImg/%06d.png, CompositeFluidAlpha/%06d.png (grey), BGImg.png
        io.save_frames(io.frames_to_uint8(outs["FluidImg"], (raw_h, raw_w)), scene, key="FluidImg")
        io.save_frames(io.alpha_to_uint8(outs["CompositeFluidAlpha"], (raw_h, raw_w)), scene, key="CompositeFluidAlpha")
        io.save_image(io.frames_to_uint8(outs["BGImg"], (raw_h, raw_w))[0], os.path.join(scene, "BGImg.png"))
    print(f"{a.N} frames at {H}x{a.W} in {dt:.2f} s ({a.N / dt:.1f} frames/s) -> {out}")
    video = io.encode_video(out, os.path.join(a.outdir, a.name, f"PredImg_{a.name}.mp4"))    # :289 (needs ffmpeg)
    print(f"video: {video}" if video else "video: skipped (no ffmpeg on PATH)")


if __name__ == "__main__":
    main()
