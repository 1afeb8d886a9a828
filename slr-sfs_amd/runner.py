"""Host side of the reference's runners, for one or several GPUs of a node.

test_animating/test_baseline_4eval_rawsize.py / test_v1_4eval_rawsize.py animate ONE scene per process on one GPU;
test_animating/CLAW/test_all_CLAW_scenes.py:70-96 walks a directory of ``<scene>_input.jpg`` + ``<scene>.flo`` pairs and
starts that script once per scene.  Here a scene is one call: the frames of its clip are rendered round-robin by the
ranks of the job (parallel.shard_frames: t = r mod G), assembled by ONE all-gather (RCCL over xGMI) and written by rank 0
-- config C5 of BASELINE.json.  With one process it is the reference's flow unchanged.  tools/animate.py (one scene) and
tools/animate_scenes.py (a directory) are the command lines."""
import os
import time

import torch

from . import io, nets, parallel, pipeline

V1_NETS = ("encoder", "projector", "net_bg", "net_alpha_encoder", "net_alpha_decoder")


def load_model(ckpt, v1, dev):
    """BaselineAnimator / SLRv1Animator on ``dev``; ``ckpt``: a reference checkpoint (``state_dict`` with
    ``model.module.<net>.`` keys, ``opts`` = the training Namespace) or None / 'None' for random-init networks.
    (Like the reference's scripts, this unpickles the file -- the Namespace is a pickled object: only load checkpoints you trust.)"""
    opts, sd = None, None
    if ckpt not in (None, "None", "none", ""):
        blob = torch.load(ckpt, map_location="cpu", weights_only=False)
        sd, opts = blob["state_dict"], blob.get("opts")
    if sd is None:
        torch.manual_seed(0)                       # random-init networks (plumbing / timing): the same on every rank
    model = pipeline.SLRv1Animator(opts=opts) if v1 else pipeline.BaselineAnimator(opts=opts)
    if sd is not None:
        for name in (V1_NETS if v1 else V1_NETS[:2]):
            nets.load_reference_state_dict(getattr(model, name), sd, "model.module." + name + ".")
    return model.to(dev).eval()


def animate_scene(model, image_path, flow_path, out_dir, name, H, W, N, speed, align=None, rank=0, world=1, group=None,
                  video=True, half_size=False):
    """One scene -> out_dir/name/PredImg/%06d.png (2-layer model: + FluidImg/, CompositeFluidAlpha/, BGImg.png;
    test_v1_4eval_rawsize.py:240-284), written by rank 0, at the raw size of the image (the *_rawsize scripts) or at half
    of it (half_size: test_baseline_4eval.py / test_v1_4eval.py:160-161).  Returns (seconds of device work, frame dir)."""
    v1 = isinstance(model, pipeline.SLRv1Animator)
    dev = next(model.parameters()).device
    image, (raw_w, raw_h) = io.load_image(image_path, H, W)
    motion = pipeline.prepare_motion(io.load_motion(flow_path), H, W, speed, io.speed_align(align, name), N)
    if half_size:
        raw_w, raw_h = raw_w // 2, raw_h // 2
    image, motion = image.to(dev), motion.to(dev)
    mine = parallel.shard_frames(N, rank, world)      # (N < world: some ranks render nothing -- they still enter every
    shard = (rank, world, group) if world > 1 else None   # collective below, with empty [0,.,H,W] contributions)
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    if v1:
        keys = pipeline.SLRv1Animator.KEYS
        outs = model.synthesize(image, motion, N, frames=mine, shard=shard, keys=keys)
        clips = {k: (v if k == "BGImg" else parallel.gather_clip(v.contiguous(), N, rank, world, group))     # BGImg: one
                 for k, v in outs.items()}                                                             # frame-invariant image
    else:
        clips = {"PredImg": parallel.gather_clip(model.synthesize(image, motion, N, frames=mine, shard=shard), N, rank, world, group)}
    torch.cuda.synchronize(dev)
    dt = time.perf_counter() - t0
    scene = os.path.join(out_dir, name)
    frame_dir = os.path.join(scene, "PredImg")
    if rank == 0:
        io.save_frames(io.frames_to_uint8(clips["PredImg"], (raw_h, raw_w)), scene)
        if v1:
            io.save_frames(io.frames_to_uint8(clips["FluidImg"], (raw_h, raw_w)), scene, key="FluidImg")
            io.save_frames(io.alpha_to_uint8(clips["CompositeFluidAlpha"], (raw_h, raw_w)), scene, key="CompositeFluidAlpha")
            io.save_image(io.frames_to_uint8(clips["BGImg"], (raw_h, raw_w))[0], os.path.join(scene, "BGImg.png"))
        if video:
            io.encode_video(frame_dir, os.path.join(scene, f"PredImg_{name}.mp4"))            # needs ffmpeg; None without
    return dt, frame_dir


def list_scenes(image_dir, flow_dir=None, align=None, start=-1, end=-1):
    """[(scene, image file, flow file)] the way test_all_CLAW_scenes.py:68-83 picks them: ``*_input.jpg`` sorted, the
    i-th kept if start <= i <= end (-1: open), scenes missing from the alignment table skipped; the flow is
    ``<scene>.flo`` next to the image (the reference ignores its flow_dir argument) or in ``flow_dir``."""
    import json
    table = None
    if align and align != "None" and os.path.exists(align):
        with open(align) as f:
            table = json.load(f)
    out = []
    images = sorted(x for x in os.listdir(image_dir) if x.endswith("_input.jpg"))
    for i, fn in enumerate(images):
        if (start != -1 and i < start) or (end != -1 and i > end):
            continue
        scene = fn[:-len("_input.jpg")]
        if table is not None and scene not in table:
            continue
        flow = os.path.join(image_dir, scene + ".flo")
        if not os.path.exists(flow) and flow_dir:
            flow = os.path.join(flow_dir, scene + ".flo")
        out.append((scene, os.path.join(image_dir, fn), flow))
    return out
