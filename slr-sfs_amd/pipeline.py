"""Clip synthesis pipelines: the build's counterpart of the reference's ``forward_flow`` +
the frame loop of its test scripts, with the frame-invariant work hoisted out of the loop.

  BaselineAnimator  <->  AnimatingSoftmaxSplating.forward_flow (models/animating_softmax_splating.py:777-981)
                         driven by test_animating/test_baseline_4eval_rawsize.py:234-274
  SLRv1Animator     <->  AnimatingSoftmaxSplatingJoint.forward_flow
                         (models/animating_softmax_splating_2layers_alpha_seperate.py:843-1108)
                         driven by test_animating/test_v1_4eval_rawsize.py:227-286

Per clip (once): encoder (and for v1: background net, alpha encoder -- the reference recomputes
the alpha encoder every frame although its input is frame-invariant, :938), Z.max(), the two
all-frames Euler passes, row lists and plans of all displacement maps.  Per batch of frames: the fused splat kernel
(HIP), the decoder(s) -- every 3x3 / 1x1 (partial) convolution, resampling and normalisation stage on this package's own
HIP kernels (csrc/conv.hip: split-f16 or fp32 matrix-core kernels; PyTorch only holds the tensors) --, tanh / compositing.
Nothing syncs the host inside the loop; frames stay on the device.
"""
import math

import torch

from . import nets
from . import parallel
from .synthesis import ClipSynthesizer, MotionPlan


_side_streams = {}


def _features_ahead(clip, frames, overlap=False):
    """Yield clip.features(t) for t in frames.

    overlap=False (default): everything on the caller's stream.
    overlap=True: frame i+1's Euler lookup + splat run on a side HIP stream while the caller's stream runs frame
    i's decoder (+1 % frames/s at 768x1280: the splat kernel itself takes 1.6x longer next to the convolutions).
    Tensors that cross streams are registered with the caching allocator (record_stream).  This mode is what
    exposed the packed-fp32 problem described in csrc/Makefile and DESIGN.md 3.2 (v_pk_fma_f32 next to a
    concurrent MFMA kernel); the library is built without those instructions and
    tests/test_gpu_parity.py::test_splat_next_to_concurrent_matrix_core_kernel keeps it that way.  (Kernels that
    are not this library's -- torch's elementwise kernels -- keep their own code generation; on the side stream
    the baseline pipeline launches none, the SLR-v1 features a few small ones.)  Off by default."""
    frames = list(frames)
    if not overlap or not frames:
        for t in frames:
            yield clip.features(t)
        return
    main = torch.cuda.current_stream()
    key = main.device.index
    if key not in _side_streams:
        _side_streams[key] = torch.cuda.Stream(device=main.device)
    side = _side_streams[key]
    side.wait_stream(main)                                          # the clip's frame-invariant tensors are ready

    def launch(t):
        with torch.cuda.stream(side):
            out = clip.features(t)
            ev = torch.cuda.Event()
            ev.record(side)
        return out, ev

    nxt = launch(frames[0])
    for i in range(len(frames)):
        out, ev = nxt
        if i + 1 < len(frames):
            nxt = launch(frames[i + 1])
        main.wait_event(ev)
        for x in (out if isinstance(out, tuple) else (out,)):
            x.record_stream(main)
        yield out
    side.wait_stream(main)


import os as _os
DECODE_BATCH = max(1, int(_os.environ.get("SLR_SFS_AMD_DECODE_BATCH", "4")))      # frames decoded per launch of the decoder networks (measured at 768x1280: 6.04 / 5.95 / 5.87 ms per
                      # frame at 1 / 2 / 4 -- fewer kernel tails; the splat writes straight into the batch buffer)


def _feature_batches(clip, frames, batch):
    """Yield (first index, gen_fs [b,C,H,W], alpha_fluid [b,1,H,W] or None) for consecutive groups of <= batch frames:
    every frame's features are written by the splat kernels directly into one sample of the batch tensors."""
    from .synthesis import MAX_BATCH
    frames = list(frames)
    H, W = clip.fs.shape[2:]
    sb = max(batch, MAX_BATCH // batch * batch)      # frames splatted together: whole decoder batches, <= MAX_BATCH
    for s0 in range(0, len(frames), sb):
        ts = frames[s0:s0 + sb]
        gen = clip.fs.new_empty(len(ts), clip.C, H, W)
        afl = clip.fs.new_empty(len(ts), 1, H, W) if clip.v1 else None
        clip.features_batch(ts, gen, afl)            # one launch of the tile kernel per weight group for (up to 8 of) them
        for j in range(0, len(ts), batch):
            yield s0 + j, gen[j:j + batch], None if afl is None else afl[j:j + batch]


def _check_grid(image):
    """The decoders go through three stride-2 poolings and three 2x up-samplings (architectures.py:345-375): on a
    grid that is not a multiple of 8 the reference's decoder returns a larger frame than it was given (and the
    2-layer compositing fails on the shape mismatch); refuse it with a message instead."""
    H, W = image.shape[2:]
    if H % 8 or W % 8:
        raise ValueError(f"working resolution {H}x{W}: height and width must be multiples of 8")


def _encode(encoder, image, shard, policy="split", owner=None, what="encoder"):
    """The per-clip networks under the convolution policy (nets.guarded).  With a shard, every rank guards ITS band before
    the all-gather: whatever a rank decides, all ranks enter the same collective."""
    guard = (lambda fn: nets.guarded(fn, image.device, policy, what, owner)) if image.is_cuda else (lambda fn: fn())
    if shard is None:
        return guard(lambda: encoder(image))
    return parallel.encode_banded(encoder, image, shard[0], shard[1], shard[2] if len(shard) > 2 else None, guard=guard)


CONV_POLICIES = ("auto", "split", "fp32", "fp32-winograd")
_RUNG_SCALE = (64.0, 1.0)


def _rung_context(rung):
    """Arithmetic of the decoder convolutions by rung of the ladder: 0 split-f16 at activation scale 2^6 (exact for
    |x| < 1023), 1 split-f16 at scale 1 (|x| < 65472), 2 the fp32 matrix instructions (nets.fp32_kernels: no limit)."""
    return nets.fp32_kernels(winograd=False) if rung >= 2 else nets.activation_scale(_RUNG_SCALE[rung])


def _render(owner, clip, frames, batch, overlap, decode, policy, on_frame, one_by_one=False):
    """The frame loop of both animators.  decode(gen, afl) -> the batch's outputs; store(pos, outputs) puts them at
    positions ``pos`` (indices into ``frames``).  ``policy`` (CONV_POLICIES):
      "fp32"  every convolution on this package's fp32 matrix-core kernels (v_mfma_f32_32x32x2_f32): the reference's arithmetic;
      "fp32-winograd"  the same rung with its 3x3 layers as Winograd F(2x2, 3x3): 1.4x the clip rate of "fp32", 2 - 4x its rounding error
              per layer (nets.fp32_kernels: what that means for whole frames);
      "split" split-f16 matrix-core kernels; an activation outside their exact range raises after the clip;
      "auto"  split-f16 kernels; every decoder batch leaves an asynchronous record of the device's saturation counter
              (no host synchronisation inside the loop); after the last batch the records are read and the batches in
              which an activation was clamped are rendered again one rung up (activation scale 1, then fp32) -- the
              clip that comes back never contains a clamped frame.  The animator remembers the rung (``_conv_rung``; see _ConvRung:
              ``reset_conv_rung()``, and rung 0 is tried again every RUNG_RETRY_CLIPS clips).
    With ``on_frame`` (frames leave the rank while the clip is still being rendered) a batch is checked before its
    frames are handed on: one host synchronisation per batch."""
    decode_batch, store = decode
    frames = list(frames)
    dev = clip.fs.device

    def groups(sub):
        if one_by_one:
            return ((i, g, a) for i, (g, a) in enumerate((x if isinstance(x, tuple) else (x, None))
                                                          for x in _features_ahead(clip, sub, overlap)))
        return _feature_batches(clip, sub, batch)

    def emit(pos):
        if on_frame is not None:
            for p_ in pos:
                on_frame(p_)

    if policy in ("fp32", "fp32-winograd") or not dev.type == "cuda":
        with (nets.fp32_kernels(winograd=policy == "fp32-winograd") if policy in ("fp32", "fp32-winograd") else _null()):
            for i0, gen, afl in groups(frames):
                pos = list(range(i0, i0 + gen.shape[0]))
                store(pos, decode_batch(gen, afl))
                emit(pos)
        return
    if policy == "split":
        for i0, gen, afl in groups(frames):
            pos = list(range(i0, i0 + gen.shape[0]))
            store(pos, decode_batch(gen, afl))
            emit(pos)
        nets.check_saturation(dev, "decoder")
        return
    assert policy == "auto", policy
    rung = getattr(owner, "_conv_rung", 0)
    if on_frame is not None:                              # frames leave as they are finished: check each batch first
        nets.saturation_count(dev)
        for i0, gen, afl in groups(frames):
            pos = list(range(i0, i0 + gen.shape[0]))
            while True:
                with _rung_context(rung):
                    outs = decode_batch(gen, afl)
                if rung >= 2 or nets.saturation_count(dev) == 0:
                    break
                rung += 1
                owner._conv_rung = rung
                _warn_rung(rung)
            store(pos, outs)
            emit(pos)
        return
    todo = list(range(len(frames)))                       # positions still to render
    while todo:
        sub = [frames[p_] for p_ in todo]
        if rung >= 2:
            with _rung_context(rung):
                for i0, gen, afl in groups(sub):
                    store([todo[i0 + k] for k in range(gen.shape[0])], decode_batch(gen, afl))
            return
        log = nets.SaturationLog(dev, len(sub))
        spans = []
        with _rung_context(rung):
            for i0, gen, afl in groups(sub):
                pos = [todo[i0 + k] for k in range(gen.shape[0])]
                store(pos, decode_batch(gen, afl))
                log.mark()
                spans.append(pos)
        bad = log.bad()
        if not bad:
            return
        todo = [p_ for b in bad for p_ in spans[b]]
        rung += 1
        owner._conv_rung = rung
        _warn_rung(rung)


RUNG_RETRY_CLIPS = 16     # convs="auto": an animator that stepped up tries rung 0 again after this many clips (0 = never)


class _ConvRung:
    """The rung of convs="auto" an animator is on (0 split-f16 at activation scale 64, 1 at scale 1, 2 fp32 kernels): a step up is
    per animator (and per rank) and shared by its encoder and decoder; it is NOT for good -- ``reset_conv_rung()`` goes back to
    rung 0 at once, and every RUNG_RETRY_CLIPS clips rendered on a raised rung the next clip starts at rung 0 again (its clamped
    batches are rendered again one rung up as always: one outlier clip does not leave a long-running service on the slow rung)."""

    def reset_conv_rung(self):
        self._conv_rung, self._rung_clips = 0, 0

    def _clip_begins(self):
        if getattr(self, "_conv_rung", 0) > 0 and RUNG_RETRY_CLIPS > 0:
            self._rung_clips = getattr(self, "_rung_clips", 0) + 1
            if self._rung_clips > RUNG_RETRY_CLIPS:
                self.reset_conv_rung()


def _warn_rung(rung):
    import warnings
    warnings.warn("slr_sfs_amd: decoder activations exceed the exact range of the split-f16 convolutions; rendering the "
                  "affected frames again " + ("at activation scale 1 (exact up to 65472)" if rung == 1 else
                                              "on the fp32 rung (v_mfma_f32_32x32x2_f32 kernels)"))


class _null:
    def __enter__(self):
        return self

    def __exit__(self, *exc):
        return False


def prepare_motion(flow, H, W, speed=1.0, align=None, N=None):
    """Motion preparation of the test scripts (test_baseline_4eval_rawsize.py:173-184,222-226):
    scale a [1,2,h,w] field to the working grid, nearest-resize it, optional speed alignment."""
    _, _, h, w = flow.shape
    flow = flow.clone()
    flow[:, 0] *= float(W) / float(w) * speed
    flow[:, 1] *= float(H) / float(h) * speed
    flow = torch.nn.functional.interpolate(flow, (H, W))            # default mode: nearest
    if align is not None:
        flow = flow * (float(align) / float(N))
    return flow.contiguous()


def _flag(opts, name, default=False):
    """``"name" in self.opt and self.opt.name`` of the reference, for an argparse.Namespace, a dict, or None."""
    if opts is None:
        return default
    if isinstance(opts, dict):
        return opts.get(name, default)
    return getattr(opts, name, default)


def _has(opts, name):
    """``"name" in self.opt``: attribute EXISTENCE on the Namespace."""
    if opts is None:
        return False
    return (name in opts) if isinstance(opts, dict) else hasattr(opts, name)


def splat_options(opts, two_layer):
    """Splat-weight options of a checkpoint's pickled ``opts`` -> ClipSynthesizer keyword arguments.

    Baseline (animating_softmax_splating.py:849-859): use_softmax_splatter_v2 / _v1, and Z_f_norm is clamped to
    [-20, 20] unless the Namespace HAS an attribute ``no_clamp_Z`` (whatever its value: checkpoints written by the
    current option parser always have it, older ones do not).  2-layer model
    (..._2layers_alpha_seperate.py:955-961): v2 / v1 only, never a clamp."""
    kw = dict(softmax_v1=bool(_flag(opts, "use_softmax_splatter_v1")) and not _flag(opts, "use_softmax_splatter_v2"),
              softmax_v2=bool(_flag(opts, "use_softmax_splatter_v2")), clamp_z=None)
    if not two_layer and opts is not None and not _has(opts, "no_clamp_Z"):
        kw["clamp_z"] = (-20.0, 20.0)
    return kw


class BaselineAnimator(torch.nn.Module, _ConvRung):
    def __init__(self, encoder=None, decoder=None, clamp_z=None, softmax_v1=False, softmax_v2=False, opts=None, convs="auto"):
        """clamp_z / softmax_v1 / softmax_v2: see ClipSynthesizer; ``opts`` (the checkpoint's pickled Namespace)
        sets them the way the reference's forward_flow reads them (splat_options).  convs: arithmetic of the encoder /
        decoder convolutions, one of CONV_POLICIES (see _render): "auto" = split-f16 matrix-core kernels with an automatic
        step up (activation scale 1, then the package's fp32 matrix-core kernels) where an activation leaves their exact range."""
        super().__init__()
        assert convs in CONV_POLICIES
        self.convs, self._conv_rung = convs, 0
        self.encoder = encoder if encoder is not None else nets.EncoderWithZ()
        self.projector = decoder if decoder is not None else nets.DecoderPconv2(64, 3)
        self.splat_kw = dict(clamp_z=clamp_z, softmax_v1=softmax_v1, softmax_v2=softmax_v2)
        if opts is not None:
            self.splat_kw = splat_options(opts, two_layer=False)

    @torch.no_grad()
    def begin_clip(self, image, motion, N, shard=None, frames=None, convs=None):
        """Frame-invariant part.  image [1,3,H,W] in [-1,1]; motion [1,2,H,W] px/frame.
        shard = (rank, world[, group]): the encoder runs in row bands across the ranks (parallel.encode_banded).
        frames: the frames that will be rendered (default all): bins / work plans are prepared for those."""
        plan = MotionPlan(motion, N, frames)                        # motion-only work first (its totals reach the host
        fs, Z = _encode(self.encoder, image, shard, convs or self.convs, self)   # under the encoder); start_fs, Z_f (:779-786)
        return ClipSynthesizer(fs, Z, motion, N, plan=plan, **self.splat_kw)

    @torch.no_grad()
    def frame(self, clip, t):
        gen_fs = clip.features(t)                                   # :847-924
        return torch.tanh(self.projector(gen_fs))                   # :973-977

    @torch.no_grad()
    def forward_flow(self, batch):
        """Reference-compatible single-frame entry (same batch keys / return dict as :777-981).
        Re-does the per-clip work on every call, like the reference; use begin_clip/frame for speed."""
        start, middle, end = [int(v) for v in torch.as_tensor(batch["index"]).reshape(-1)[:3]]
        fs, Z = batch["features"][0][:2]
        clip = ClipSynthesizer(fs, Z.view(fs.shape[0], 1, fs.shape[2], fs.shape[3]), batch["motions"][0],
                               end - start + 1, frames=[middle - start], **self.splat_kw)
        gen = clip.features(middle - start)
        return {"PredImg": torch.tanh(self.projector(gen)), "Z_f": Z}

    @torch.no_grad()
    def synthesize(self, image, motion, N, frames=None, overlap=False, on_frame=None, shard=None, batch=None, convs=None):
        """All (or the given) frames of one clip -> [len(frames),3,H,W] on the device.  batch (default DECODE_BATCH):
        frames per decoder launch; overlap=True (see _features_ahead) decodes frame by frame; convs: overrides the
        animator's convolution policy for this clip (CONV_POLICIES)."""
        _check_grid(image)
        frames = list(range(N) if frames is None else frames)
        policy = self.convs if convs is None else convs
        assert policy in CONV_POLICIES
        self._clip_begins()
        clip = self.begin_clip(image, motion, N, shard, frames, convs=policy)
        out = image.new_empty(len(frames), 3, image.shape[2], image.shape[3])
        batch = DECODE_BATCH if batch is None else max(1, int(batch))

        def store(pos, raw):                              # raw = the projector's output: the tanh writes the frames where they belong
            if pos and pos[-1] - pos[0] + 1 == len(pos):   # (no intermediate tensor, no 47 MB device copy per batch)
                torch.tanh(raw, out=out[pos[0]:pos[0] + len(pos)])
            else:
                out[torch.as_tensor(pos, device=out.device)] = torch.tanh(raw)

        _render(self, clip, frames, batch, overlap, (lambda gen, afl: self.projector(gen), store), policy,
                None if on_frame is None else (lambda p_: on_frame(out[p_])), one_by_one=overlap or batch == 1)
        return out


def blur_alpha_region(alpha_region, W):
    """The edit mask of ..._2layers_alpha_seperate.py:867-906: Gaussian blur with a (W // 20, made odd)-wide kernel,
    sigma = W // 50, replicate padding.  An optional editing feature, once per clip: plain torch."""
    k = W // 20
    if k % 2 == 0:
        k = k + 1
    sigma = W // 50
    xc = torch.arange(k)
    xg = xc.repeat(k).view(k, k)
    xy = torch.stack([xg, xg.t()], dim=-1)
    mean, var = (k - 1) / 2.0, float(sigma ** 2)
    g = (1.0 / (2.0 * math.pi * var)) * torch.exp(-torch.sum((xy - mean) ** 2.0, dim=-1).float() / (2.0 * var))
    g = (g / torch.sum(g)).view(1, 1, k, k).to(alpha_region.device)
    x = torch.nn.functional.pad(alpha_region, (k // 2,) * 4, mode="replicate")
    return torch.nn.functional.conv2d(x, g)


class SLRv1Animator(torch.nn.Module, _ConvRung):
    KEYS = ("PredImg", "BGImg", "FluidImg", "CompositeFluidAlpha")

    def __init__(self, encoder=None, decoder=None, net_bg=None, alpha_encoder=None, alpha_decoder=None,
                 use_alpha0=True, softmax_v1=False, softmax_v2=False, use_alpha_softmax=False, clamp_alpha=0.0,
                 use_fluid_alpha_only=False, use_bg_alpha_only=False, opts=None, convs="auto"):
        """Compositing options of ..._2layers_alpha_seperate.py:1060-1085 (all off in the shipped scripts):
        use_alpha_softmax, clamp_alpha (> 0: lower bound of the composited fluid alpha -- not the clamp of the time
        weight, which this model always applies, :952), use_fluid_alpha_only, use_bg_alpha_only.  ``opts`` (the
        checkpoint's pickled Namespace) sets all of them, use_alpha0 and the splat-weight variant."""
        super().__init__()
        assert convs in CONV_POLICIES
        self.convs, self._conv_rung = convs, 0                       # (see BaselineAnimator / _render)
        self.encoder = encoder if encoder is not None else nets.EncoderWithZ()
        self.projector = decoder if decoder is not None else nets.DecoderPconv2(64, 3)
        self.net_bg = net_bg if net_bg is not None else nets.BGDecoder()
        self.net_alpha_encoder = alpha_encoder if alpha_encoder is not None else nets.Encoder(3, 2)
        self.net_alpha_decoder = alpha_decoder if alpha_decoder is not None else nets.DecoderPconv2(65, 1)
        self.use_alpha0 = use_alpha0
        self.splat_kw = dict(softmax_v1=softmax_v1, softmax_v2=softmax_v2)
        self.use_alpha_softmax, self.clamp_alpha = bool(use_alpha_softmax), float(clamp_alpha)
        self.use_fluid_alpha_only, self.use_bg_alpha_only = bool(use_fluid_alpha_only), bool(use_bg_alpha_only)
        if opts is not None:
            kw = splat_options(opts, two_layer=True)
            self.splat_kw = dict(softmax_v1=kw["softmax_v1"], softmax_v2=kw["softmax_v2"])
            self.use_alpha0 = bool(_flag(opts, "use_alpha0_as_blending_weight"))
            self.use_alpha_softmax = bool(_flag(opts, "use_alpha_softmax"))
            self.clamp_alpha = float(_flag(opts, "clamp_alpha", 0.0) or 0.0)
            self.use_fluid_alpha_only = bool(_flag(opts, "use_fluid_alpha_only"))
            self.use_bg_alpha_only = bool(_flag(opts, "use_bg_alpha_only"))

    def _clip(self, fs, Z, motion, N, alpha_out, bg, alpha_region=None, plan=None, frames=None):
        alpha_bg_raw = alpha_out[:, 0:1]                            # :943-946
        alpha_bg = torch.sigmoid(alpha_bg_raw)
        clip = ClipSynthesizer(fs, Z, motion, N, alpha_fluid_logit=alpha_out[:, 1:2].contiguous(), alpha_bg=alpha_bg,
                               use_alpha0=self.use_alpha0, plan=plan, frames=frames, **self.splat_kw)
        clip.bg, clip.alpha_bg, clip.alpha_bg_raw = bg, alpha_bg, alpha_bg_raw
        clip.alpha_region = None if alpha_region is None else blur_alpha_region(alpha_region, fs.shape[3])
        return clip

    @torch.no_grad()
    def begin_clip(self, image, motion, N, shard=None, alpha_region=None, frames=None, convs=None):
        """alpha_region [1,1,H,W]: optional edit mask (:867-906, 1079-1080): 1 = composite, 0 = fluid layer only."""
        policy = convs or self.convs
        plan = MotionPlan(motion, N, frames)
        fs, Z = _encode(self.encoder, image, shard, policy, self)
        bg = torch.tanh(_encode(self.net_bg, image, None, policy, self, "background network"))   # test_v1_4eval_rawsize.py:209, :925-927
        a = _encode(self.net_alpha_encoder, image, shard, policy, self, "alpha encoder")         # :938 (frame-invariant -> hoisted)
        return self._clip(fs, Z, motion, N, a, bg, alpha_region, plan=plan)

    @torch.no_grad()
    def frame(self, clip, t):
        return self._decode(clip, *clip.features(t))                # :950-1045

    def _decode(self, clip, gen_fs, alpha_fluid):
        fluid = torch.tanh(self.projector(gen_fs))                  # :1048-1049
        fa_raw = self.net_alpha_decoder(torch.cat([gen_fs, alpha_fluid], 1))                       # :1052-1053
        fluid_alpha = torch.sigmoid(fa_raw)                                                        # :1054
        alpha_bg, bg = clip.alpha_bg, clip.bg
        alpha_norm = torch.clamp(fluid_alpha + alpha_bg, min=1e-8)                                 # :1056-1057
        if self.use_fluid_alpha_only or self.use_bg_alpha_only:                                    # :1060-1063
            alpha_norm = 1
        if self.use_alpha_softmax:                                                                 # :1066-1070
            ca = torch.softmax(torch.cat([fa_raw, clip.alpha_bg_raw], 1), dim=1)
            pred = ca[:, :1] * fluid + ca[:, 1:2] * bg
        elif self.clamp_alpha > 0:                                                                 # :1071-1075
            cfa = torch.clamp(fluid_alpha / alpha_norm, min=self.clamp_alpha)
            pred = cfa * fluid + (1.0 - cfa) * bg
        else:
            pred = (fluid_alpha * fluid + alpha_bg * bg) / alpha_norm                              # :1077
        region = getattr(clip, "alpha_region", None)
        if region is not None:                                                                     # :1079-1080
            pred = pred * region + fluid * (1.0 - region)
        if self.use_fluid_alpha_only:                                                              # :1082-1083
            pred = fluid_alpha * fluid + (1.0 - fluid_alpha) * bg
        if self.use_bg_alpha_only:                                                                 # :1084-1085
            pred = (1.0 - alpha_bg) * fluid + alpha_bg * bg
        cfa = fluid_alpha / alpha_norm                                                             # :1088
        out = {"PredImg": pred, "BGImg": bg, "FluidImg": fluid, "CompositeFluidAlpha": cfa}        # :1089-1093
        if region is not None:                                                                     # :1100-1103
            out["EditedCompositeFluidAlpha"] = cfa * region + 1.0 * (1.0 - region)
            out["AlphaRegionMask"] = region
        if self.use_fluid_alpha_only:
            # :1104-1107 -- upstream assigns gen_fluid_alpha and then, under
            # `"use_bg_alpha_only" in self.opt and self.opt.use_fluid_alpha_only` (the option EXISTS in every parsed
            # Namespace, and the value tested is use_FLUID_alpha_only), overwrites it with alpha_bg_f.  Kept as it is
            # (tests/golden/pipeline_v1_surface.npz holds the reference's output).
            out["CompositeFluidAlpha"] = alpha_bg
        return out

    @torch.no_grad()
    def forward_flow(self, batch):
        """Reference-compatible single-frame entry: same batch keys and return dict as
        AnimatingSoftmaxSplatingJoint.forward_flow (..._2layers_alpha_seperate.py:843-1108), as the runner calls it
        (test_v1_4eval_rawsize.py:227-240): batch["features"] = [(start_fs, Z_f)], batch["BGImg"] = [net_bg(img)]
        (before the tanh, :925-927), batch["images"], batch["motions"], batch["index"] = [[0, t, N-1]], optional
        batch["alpha_region"].  Like the reference it re-runs the alpha encoder and the Euler integration on every
        call; begin_clip / frame hoist them."""
        idx = torch.as_tensor(batch["index"])
        start, middle, end = [int(v) for v in idx.reshape(-1, 3)[0]]
        assert idx.numel() == 3, "one frame per call (the reference's euler_integration is batch-1 only as well)"
        image = batch["images"][0]
        fs, Z = batch["features"][0][:2]
        assert fs.shape[0] == 1
        a = self.net_alpha_encoder(image)                           # :938
        bg = torch.tanh(batch["BGImg"][0])                          # :925-927
        clip = self._clip(fs, Z.view(1, 1, fs.shape[2], fs.shape[3]), batch["motions"][0], end - start + 1, a, bg,
                          batch.get("alpha_region"), frames=[middle - start])
        return self._decode(clip, *clip.features(middle - start))

    @torch.no_grad()
    def synthesize(self, image, motion, N, frames=None, overlap=False, on_frame=None, shard=None, keys=None,
                   alpha_region=None, batch=None, convs=None):
        """keys=None: PredImg frames [n,3,H,W] (as BaselineAnimator.synthesize).  keys=("PredImg", "FluidImg",
        "CompositeFluidAlpha", "BGImg", ...): a dict of those outputs of forward_flow, stacked over the frames
        ("BGImg" and "AlphaRegionMask" are frame-invariant: one [1,.,H,W] tensor) -- what
        test_v1_4eval_rawsize.py:240-284 writes to disk.  An empty ``frames`` (a rank without frames of a short clip)
        returns empty [0,.,H,W] tensors for every key, so that the callers' collectives are entered by every rank."""
        _check_grid(image)
        frames = list(range(N) if frames is None else frames)
        policy = self.convs if convs is None else convs
        assert policy in CONV_POLICIES
        self._clip_begins()
        want = ("PredImg",) if keys is None else tuple(keys)
        once = ("BGImg", "AlphaRegionMask")
        channels = {"PredImg": 3, "FluidImg": 3, "CompositeFluidAlpha": 1, "EditedCompositeFluidAlpha": 1}
        # every key is checked BEFORE any device work (and before any collective a sharded caller will enter)
        for k in want:
            if k not in channels and k not in once:
                raise KeyError(f"SLRv1Animator.synthesize: unknown output {k!r}")
            if k in ("EditedCompositeFluidAlpha", "AlphaRegionMask") and alpha_region is None:
                raise KeyError(f"SLRv1Animator.synthesize: {k!r} needs an alpha_region")
        if on_frame is not None and "PredImg" not in want:
            raise KeyError("SLRv1Animator.synthesize: on_frame receives PredImg frames: request that key")
        clip = self.begin_clip(image, motion, N, shard, alpha_region, frames, convs=policy)
        H, W = image.shape[2:]
        outs = {k: image.new_empty(len(frames), channels[k], H, W) for k in want if k not in once}
        if "BGImg" in want:
            outs["BGImg"] = clip.bg
        if "AlphaRegionMask" in want:
            outs["AlphaRegionMask"] = clip.alpha_region
        batch = DECODE_BATCH if batch is None else max(1, int(batch))

        def store(pos, d):
            idx = None if (pos and pos[-1] - pos[0] + 1 == len(pos)) else torch.as_tensor(pos, device=image.device)
            for k in want:
                if k in once:
                    continue
                if idx is None:
                    outs[k][pos[0]:pos[0] + len(pos)] = d[k]
                else:
                    outs[k][idx] = d[k]

        _render(self, clip, frames, batch, overlap, (lambda gen, afl: self._decode(clip, gen, afl), store), policy,
                None if on_frame is None else (lambda p_: on_frame(outs["PredImg"][p_])),
                one_by_one=overlap or batch == 1 or not clip.use_alpha0)
        return outs["PredImg"] if keys is None else outs
