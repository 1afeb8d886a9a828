"""Clip synthesis pipelines: the build's counterpart of the reference's ``forward_flow`` +
the frame loop of its test scripts, with the frame-invariant work hoisted out of the loop.

  BaselineAnimator  <->  AnimatingSoftmaxSplating.forward_flow (models/animating_softmax_splating.py:777-981)
                         driven by test_animating/test_baseline_4eval_rawsize.py:234-274
  SLRv1Animator     <->  AnimatingSoftmaxSplatingJoint.forward_flow
                         (models/animating_softmax_splating_2layers_alpha_seperate.py:843-1108)
                         driven by test_animating/test_v1_4eval_rawsize.py:227-286

Per clip (once): encoder (and for v1: background net, alpha encoder -- the reference recomputes
the alpha encoder every frame although its input is frame-invariant, :938), Z.max(), the two
all-frames Euler passes.  Per frame: bin + fused splat (HIP), decoder(s) (PyTorch-ROCm), tanh /
compositing.  Nothing syncs the host inside the loop; frames stay on the device.
"""
import torch

from . import nets
from . import parallel
from .synthesis import ClipSynthesizer


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


def _check_grid(image):
    """The decoders go through three stride-2 poolings and three 2x up-samplings (architectures.py:345-375): on a
    grid that is not a multiple of 8 the reference's decoder returns a larger frame than it was given (and the
    2-layer compositing fails on the shape mismatch); refuse it with a message instead."""
    H, W = image.shape[2:]
    if H % 8 or W % 8:
        raise ValueError(f"working resolution {H}x{W}: height and width must be multiples of 8")


def _encode(encoder, image, shard):
    if shard is None:
        return encoder(image)
    return parallel.encode_banded(encoder, image, shard[0], shard[1], shard[2] if len(shard) > 2 else None)


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


class BaselineAnimator(torch.nn.Module):
    def __init__(self, encoder=None, decoder=None):
        super().__init__()
        self.encoder = encoder if encoder is not None else nets.EncoderWithZ()
        self.projector = decoder if decoder is not None else nets.DecoderPconv2(64, 3)

    @torch.no_grad()
    def begin_clip(self, image, motion, N, shard=None):
        """Frame-invariant part.  image [1,3,H,W] in [-1,1]; motion [1,2,H,W] px/frame.
        shard = (rank, world[, group]): the encoder runs in row bands across the ranks (parallel.encode_banded)."""
        fs, Z = _encode(self.encoder, image, shard)                 # start_fs, Z_f (:779-786)
        return ClipSynthesizer(fs, Z, motion, N)

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
                               end - start + 1)
        gen = clip.features(middle - start)
        return {"PredImg": torch.tanh(self.projector(gen)), "Z_f": Z}

    @torch.no_grad()
    def synthesize(self, image, motion, N, frames=None, overlap=False, on_frame=None, shard=None):
        """All (or the given) frames of one clip -> [len(frames),3,H,W] on the device (overlap: see _features_ahead)."""
        _check_grid(image)
        clip = self.begin_clip(image, motion, N, shard)
        frames = range(N) if frames is None else frames
        out = image.new_empty(len(frames), 3, image.shape[2], image.shape[3])
        for i, gen_fs in enumerate(_features_ahead(clip, frames, overlap)):
            out[i] = torch.tanh(self.projector(gen_fs))[0]
            if on_frame is not None:
                on_frame(out[i])                                    # e.g. parallel.ClipAssembler.push
        return out


class SLRv1Animator(torch.nn.Module):
    def __init__(self, encoder=None, decoder=None, net_bg=None, alpha_encoder=None, alpha_decoder=None,
                 use_alpha0=True):
        super().__init__()
        self.encoder = encoder if encoder is not None else nets.EncoderWithZ()
        self.projector = decoder if decoder is not None else nets.DecoderPconv2(64, 3)
        self.net_bg = net_bg if net_bg is not None else nets.BGDecoder()
        self.net_alpha_encoder = alpha_encoder if alpha_encoder is not None else nets.Encoder(3, 2)
        self.net_alpha_decoder = alpha_decoder if alpha_decoder is not None else nets.DecoderPconv2(65, 1)
        self.use_alpha0 = use_alpha0

    @torch.no_grad()
    def begin_clip(self, image, motion, N, shard=None):
        fs, Z = _encode(self.encoder, image, shard)
        bg = torch.tanh(self.net_bg(image))                         # test_v1_4eval_rawsize.py:209, :925-927
        a = _encode(self.net_alpha_encoder, image, shard)           # :938 (frame-invariant -> hoisted)
        alpha_bg = torch.sigmoid(a[:, 0:1])                         # :943-946
        clip = ClipSynthesizer(fs, Z, motion, N, alpha_fluid_logit=a[:, 1:2].contiguous(), alpha_bg=alpha_bg,
                               use_alpha0=self.use_alpha0)
        clip.bg, clip.alpha_bg = bg, alpha_bg
        return clip

    @torch.no_grad()
    def frame(self, clip, t):
        return self._decode(clip, *clip.features(t))                # :950-1045

    def _decode(self, clip, gen_fs, alpha_fluid):
        fluid = torch.tanh(self.projector(gen_fs))                  # :1048-1049
        fluid_alpha = torch.sigmoid(self.net_alpha_decoder(torch.cat([gen_fs, alpha_fluid], 1)))   # :1052-1054
        alpha_norm = torch.clamp(fluid_alpha + clip.alpha_bg, min=1e-8)                            # :1056-1057
        pred = (fluid_alpha * fluid + clip.alpha_bg * clip.bg) / alpha_norm                        # :1077
        return {"PredImg": pred, "BGImg": clip.bg, "FluidImg": fluid,
                "CompositeFluidAlpha": fluid_alpha / alpha_norm}                                   # :1087-1093

    @torch.no_grad()
    def synthesize(self, image, motion, N, frames=None, overlap=False, on_frame=None, shard=None):
        _check_grid(image)
        clip = self.begin_clip(image, motion, N, shard)
        frames = range(N) if frames is None else frames
        out = image.new_empty(len(frames), 3, image.shape[2], image.shape[3])
        for i, (gen_fs, alpha_fluid) in enumerate(_features_ahead(clip, frames, overlap)):
            out[i] = self._decode(clip, gen_fs, alpha_fluid)["PredImg"][0]
            if on_frame is not None:
                on_frame(out[i])
        return out
