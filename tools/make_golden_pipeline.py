"""P1 fixtures: the tensor the reference's own ``forward_flow`` hands to its decoder(s).

Runs the UNMODIFIED reference methods
  AnimatingSoftmaxSplating.forward_flow        (models/animating_softmax_splating.py:777-981)
  AnimatingSoftmaxSplatingJoint.forward_flow   (models/animating_softmax_splating_2layers_alpha_seperate.py:843-1108)
as unbound functions on a stand-in ``self`` that carries the reference's parsed options
(options/train_options.py, canonical flag sets of train_animating_scripts/*.sh), the reference's
``softsplat.ModuleSoftsplat('summation')`` and recording stubs in place of the decoders (and a
fixed random map in place of the alpha encoder).  What is recorded is the decoder INPUT --
i.e. everything between the encoder output and the decoder: Euler step counts, alpha, exp
weighting, channel packing, two-direction accumulation and normalisation.

Called from tools/make_golden.py (needs /root/reference; only numbers are committed).
"""
import os
import sys
import types

import numpy as np
import torch


def _stub(name, **kw):
    m = types.ModuleType(name)
    m.__dict__.update(kw)
    sys.modules[name] = m
    return m


class _Recorder(torch.nn.Module):
    def __init__(self, out_ch):
        super().__init__()
        self.out_ch = out_ch
        self.seen = []

    def forward(self, x):
        self.seen.append(x.detach().clone())
        return x.new_zeros(x.shape[0], self.out_ch, x.shape[2], x.shape[3])


class _Fixed(torch.nn.Module):
    def __init__(self, value):
        super().__init__()
        self.value = value

    def forward(self, x):
        return self.value


def _plain(t):
    return t.detach().as_subclass(torch.Tensor).numpy().astype(np.float32)


def capture_pipeline(ss, eim, out_dir, rng, cudalike):
    for n in ("cv2", "av", "lz4framed"):
        _stub(n)
    tv = _stub("torchvision")
    tv.transforms = _stub("torchvision.transforms")
    tv.models = _stub("torchvision.models", vgg19=None)
    tv.utils = _stub("torchvision.utils")
    torch.Tensor.cuda = lambda self, *a, **k: self

    import models.animating_softmax_splating as A
    import models.animating_softmax_splating_2layers_alpha_seperate as B
    # both model files did `from ...euler_integration_manipulator import euler_integration`;
    # the function object is the reference's own, its module global torch is the CPU proxy.
    from options.train_options import ArgumentParser

    W, N = 20, 60
    base = ("--model_type softmax_splating --refine_model_type resnet_256W8UpDown64_de_resnet_pconv2_nonorm "
            "--pconv pconv_pbn_woresbias --norm_G sync:spectral_batch --train_Z --use_softmax_splatter "
            "--losses 1.0_l1 --W %d" % W)
    v1 = base.replace("softmax_splating ", "softmax_splating_2layers_alpha_seperate ") + \
        (" --bg_refine_model_type resnet_256W8UpDown64BG_nonorm "
         "--alpha_refine_model_type resnet_256W8UpDown64Layers_de_resnet_pconv2_nonorm "
         "--out_channel 65 --ngf 64 --train_bg --train_alpha --use_alpha0_as_blending_weight")
    opt_base, _ = ArgumentParser().parse(base)
    opt_v1, _ = ArgumentParser().parse(v1)
    opt_v1_noa0, _ = ArgumentParser().parse(v1.replace(" --use_alpha0_as_blending_weight", ""))

    y, x = np.meshgrid(np.arange(W, dtype=np.float32), np.arange(W, dtype=np.float32), indexing="ij")
    u = 1.5 * np.sin(2 * np.pi * (2 * x / W + y / W) + 0.3)
    v = 1.5 * np.cos(2 * np.pi * (x / W - 1.5 * y / W) + 1.1)
    m = (x >= 0.35 * W).astype(np.float32)
    motion = np.stack([u * m, v * m])[None].astype(np.float32)
    fs = rng.standard_normal((1, 64, W, W)).astype(np.float32)
    Z = (rng.standard_normal((1, 1, W, W)) * 3).astype(np.float32)
    img = rng.uniform(-1, 1, (1, 3, W, W)).astype(np.float32)
    alpha_out = rng.standard_normal((1, 2, W, W)).astype(np.float32)      # [bg logit, fluid logit]
    bg = rng.standard_normal((1, 3, W, W)).astype(np.float32)

    g = {"fs": fs, "Z": Z, "motion": motion, "img": img, "alpha_out": alpha_out, "bg": bg,
         "N": np.int32(N), "ts": np.array([0, 1, 30, 59], np.int32)}

    for t in (0, 1, 30, 59):
        batch = {"features": [(cudalike(fs), cudalike(Z))], "images": [cudalike(img)],
                 "motions": [cudalike(motion)], "index": torch.tensor([[0, t, N - 1]])}
        # ---- baseline
        rec = _Recorder(3)
        me = types.SimpleNamespace(opt=opt_base, softsplater=ss.ModuleSoftsplat("summation"), projector=rec)
        pred = A.AnimatingSoftmaxSplating.forward_flow(me, batch)
        assert pred["PredImg"].shape == (1, 3, W, W)
        g[f"baseline_t{t}_gen_fs"] = _plain(rec.seen[0])
        # ---- v1 with / without use_alpha0_as_blending_weight
        for tag, opt in (("v1", opt_v1), ("v1noa0", opt_v1_noa0)):
            if tag == "v1noa0" and t not in (1, 30):
                continue
            rec_p, rec_a = _Recorder(3), _Recorder(1)
            me = types.SimpleNamespace(opt=opt, softsplater=ss.ModuleSoftsplat("summation"),
                                       projector=rec_p, net_alpha_decoder=rec_a,
                                       net_alpha_encoder=_Fixed(cudalike(alpha_out)))
            b = dict(batch)
            b["BGImg"] = [cudalike(bg)]
            pred = B.AnimatingSoftmaxSplatingJoint.forward_flow(me, b)
            g[f"{tag}_t{t}_gen_fs"] = _plain(rec_p.seen[0])
            g[f"{tag}_t{t}_dec_alpha_in"] = _plain(rec_a.seen[0])          # cat[gen_fs, alpha_fluid]
            # with zero decoder outputs: fluid=tanh(0)=0, alpha=sigmoid(0)=.5 -> exercises compositing
            g[f"{tag}_t{t}_PredImg"] = _plain(pred["PredImg"])
            g[f"{tag}_t{t}_CompositeFluidAlpha"] = _plain(pred["CompositeFluidAlpha"])
    np.savez_compressed(os.path.join(out_dir, "pipeline_a6.npz"), **g)


def a6_large_inputs(H=768, W=1280):
    """Seeded inputs of the full-size a6 digests (regenerated identically by tests/conftest.py::a6_large_case and
    by bench.py's parity check; only digests of the REFERENCE's outputs are stored)."""
    rng = np.random.default_rng(2000 + H)
    y, x = np.meshgrid(np.arange(H, dtype=np.float32), np.arange(W, dtype=np.float32), indexing="ij")
    u = 1.5 * np.sin(2 * np.pi * (2 * x / W + y / H) + 0.9)
    v = 1.5 * np.cos(2 * np.pi * (x / W - 1.5 * y / H) + 0.4)
    m = (x >= 0.35 * W).astype(np.float32)
    motion = np.stack([u * m, v * m])[None].astype(np.float32)
    fs = rng.standard_normal((1, 64, H, W)).astype(np.float32)
    Z = rng.standard_normal((1, 1, H, W)).astype(np.float32)
    alpha_out = rng.standard_normal((1, 2, H, W)).astype(np.float32)
    return fs, Z, motion, alpha_out


def capture_pipeline_large(ss, out_dir, cudalike):
    """L2 fixtures: digests of the decoder input of the reference's own forward_flow (baseline and SLR v1),
    64 features, N = 60, t in {1, 30, 59}, on two grids:
      sq : 768x768  -- the reference's real working grid (test_baseline_4eval_rawsize.py:99,165), run as it is;
      c3 : 768x1280 -- the grid BASELINE.json quotes and bench.py times.  The reference's forward_flow assumes a
           square grid in three reshapes (Z_f.view(bs,1,W,W) animating_softmax_splating.py:786, and
           forward_flow.view(bs,-1,W,W) ..._2layers_alpha_seperate.py:988,1025); for this grid Z and the motion are
           handed over as a tensor subclass whose .view() to (..., W, W) keeps the [.,.,768,1280] shape they
           already have.  Every arithmetic statement of the reference runs unmodified.
    Must run after capture_pipeline (module stubs)."""
    import models.animating_softmax_splating as A
    import models.animating_softmax_splating_2layers_alpha_seperate as B
    from options.train_options import ArgumentParser
    N, SQ = 60, 768
    base = ("--model_type softmax_splating --refine_model_type resnet_256W8UpDown64_de_resnet_pconv2_nonorm "
            "--pconv pconv_pbn_woresbias --norm_G sync:spectral_batch --train_Z --use_softmax_splatter "
            "--losses 1.0_l1 --W %d" % SQ)
    v1 = base.replace("softmax_splating ", "softmax_splating_2layers_alpha_seperate ") + \
        (" --bg_refine_model_type resnet_256W8UpDown64BG_nonorm "
         "--alpha_refine_model_type resnet_256W8UpDown64Layers_de_resnet_pconv2_nonorm "
         "--out_channel 65 --ngf 64 --train_bg --train_alpha --use_alpha0_as_blending_weight")
    opt_base, _ = ArgumentParser().parse(base)
    opt_v1, _ = ArgumentParser().parse(v1)
    base_cls = type(cudalike(np.zeros(1, np.float32)))

    class KeepGrid(base_cls):
        def view(self, *shape):
            shape = tuple(shape[0]) if len(shape) == 1 and isinstance(shape[0], (tuple, list)) else shape
            if len(shape) == 4 and tuple(shape[-2:]) == (SQ, SQ) and self.dim() == 4 and self.shape[-2:] != (SQ, SQ):
                return self
            return super().view(*shape)

    pos_rng = np.random.default_rng(77)
    g = {"N": np.int32(N), "ts": np.array([1, 30, 59], np.int32)}
    for tag, (H, W) in {"sq": (SQ, SQ), "c3": (768, 1280)}.items():
        keep = (lambda a: cudalike(a).as_subclass(KeepGrid)) if W != H else cudalike
        fs, Z, motion, alpha_out = a6_large_inputs(H, W)
        img = np.zeros((1, 3, H, W), np.float32)
        g[f"{tag}_shape"] = np.array([1, 64, H, W], np.int32)
        pos = g[f"{tag}_pos"] = pos_rng.integers(0, 64 * H * W, 4096)
        apos = g[f"{tag}_apos"] = pos_rng.integers(0, H * W, 4096)
        for t in (1, 30, 59):
            batch = {"features": [(cudalike(fs), keep(Z))], "images": [cudalike(img)],
                     "motions": [keep(motion)], "index": torch.tensor([[0, t, N - 1]])}
            rec = _Recorder(3)
            me = types.SimpleNamespace(opt=opt_base, softsplater=ss.ModuleSoftsplat("summation"), projector=rec)
            A.AnimatingSoftmaxSplating.forward_flow(me, batch)
            gen = _plain(rec.seen[0])
            assert gen.shape == (1, 64, H, W)
            g[f"{tag}_baseline_t{t}_val"] = gen.ravel()[pos]
            g[f"{tag}_baseline_t{t}_sum"] = gen.astype(np.float64).sum(axis=(2, 3))
            g[f"{tag}_baseline_t{t}_holes"] = np.int64((gen == 0).sum())
            rec_p, rec_a = _Recorder(3), _Recorder(1)
            me = types.SimpleNamespace(opt=opt_v1, softsplater=ss.ModuleSoftsplat("summation"), projector=rec_p,
                                       net_alpha_decoder=rec_a, net_alpha_encoder=_Fixed(cudalike(alpha_out)))
            b = dict(batch)
            b["BGImg"] = [cudalike(img)]
            B.AnimatingSoftmaxSplatingJoint.forward_flow(me, b)
            gen, ain = _plain(rec_p.seen[0]), _plain(rec_a.seen[0])
            assert ain.shape == (1, 65, H, W)
            g[f"{tag}_v1_t{t}_val"] = gen.ravel()[pos]
            g[f"{tag}_v1_t{t}_sum"] = gen.astype(np.float64).sum(axis=(2, 3))
            g[f"{tag}_v1_t{t}_holes"] = np.int64((gen == 0).sum())
            g[f"{tag}_v1_t{t}_alpha_val"] = ain[0, 64].ravel()[apos]
            g[f"{tag}_v1_t{t}_alpha_sum"] = np.float64(ain[0, 64].astype(np.float64).sum())
            print(f"  a6 large {tag} t={t}: baseline holes {int(g[f'{tag}_baseline_t{t}_holes'])}", flush=True)
    np.savez_compressed(os.path.join(out_dir, "pipeline_a6_large.npz"), **g)


def v1_surface_inputs(W=60):
    """Seeded inputs of tests/golden/pipeline_v1_surface.npz (regenerated by tests/conftest.py::v1_surface_inputs)."""
    rng = np.random.default_rng(4100 + W)
    y, x = np.meshgrid(np.arange(W, dtype=np.float32), np.arange(W, dtype=np.float32), indexing="ij")
    u = 1.2 * np.sin(2 * np.pi * (2 * x / W + y / W) + 0.2)
    v = 1.2 * np.cos(2 * np.pi * (x / W - 1.5 * y / W) + 1.3)
    motion = np.stack([u, v])[None].astype(np.float32)
    d = {"motion": motion,
         "fs": rng.standard_normal((1, 64, W, W)).astype(np.float32),
         "Z": (rng.standard_normal((1, 1, W, W)) * 2).astype(np.float32),
         "img": rng.uniform(-1, 1, (1, 3, W, W)).astype(np.float32),
         "alpha_out": rng.standard_normal((1, 2, W, W)).astype(np.float32),
         "bg_raw": rng.standard_normal((1, 3, W, W)).astype(np.float32),
         "dec_out": rng.standard_normal((1, 3, W, W)).astype(np.float32),          # stand-in decoder outputs
         "adec_out": (rng.standard_normal((1, 1, W, W)) * 2).astype(np.float32)}
    region = np.zeros((1, 1, W, W), np.float32)
    region[0, 0, W // 4: 3 * W // 4, W // 3:] = 1.0
    d["alpha_region"] = region
    return d


def capture_v1_surface(ss, out_dir, cudalike):
    """P2 fixtures: the full return dict of the reference's SLR-v1 forward_flow
    (..._2layers_alpha_seperate.py:843-1108) with the optional compositing paths switched on one at a time --
    alpha_region (:867-906,1079-1080,1100-1103), clamp_alpha, use_alpha_softmax, use_fluid_alpha_only,
    use_bg_alpha_only (:1060-1085,1104-1107) -- and fixed random maps in place of the two decoders and the alpha
    encoder.  Must run after capture_pipeline (module stubs)."""
    import models.animating_softmax_splating_2layers_alpha_seperate as B
    from options.train_options import ArgumentParser
    W, N, t = 60, 24, 9
    v1 = ("--model_type softmax_splating_2layers_alpha_seperate --refine_model_type resnet_256W8UpDown64_de_resnet_pconv2_nonorm "
          "--pconv pconv_pbn_woresbias --norm_G sync:spectral_batch --train_Z --use_softmax_splatter "
          "--losses 1.0_l1 --W %d --bg_refine_model_type resnet_256W8UpDown64BG_nonorm "
          "--alpha_refine_model_type resnet_256W8UpDown64Layers_de_resnet_pconv2_nonorm "
          "--out_channel 65 --ngf 64 --train_bg --train_alpha --use_alpha0_as_blending_weight" % W)
    d = v1_surface_inputs(W)
    variants = {"plain": "", "region": "", "clamp": " --clamp_alpha 0.6", "softmax": " --use_alpha_softmax",
                "fluidonly": " --use_fluid_alpha_only", "bgonly": " --use_bg_alpha_only",
                "v1weights": " --use_softmax_splatter_v1"}
    g = {"W": np.int32(W), "N": np.int32(N), "t": np.int32(t)}
    for tag, flags in variants.items():
        opt, _ = ArgumentParser().parse(v1 + flags)
        me = types.SimpleNamespace(opt=opt, softsplater=ss.ModuleSoftsplat("summation"),
                                   projector=_Fixed(cudalike(d["dec_out"])),
                                   net_alpha_decoder=_Fixed(cudalike(d["adec_out"])),
                                   net_alpha_encoder=_Fixed(cudalike(d["alpha_out"])))
        batch = {"features": [(cudalike(d["fs"]), cudalike(d["Z"]))], "images": [cudalike(d["img"])],
                 "motions": [cudalike(d["motion"])], "index": torch.tensor([[0, t, N - 1]]),
                 "BGImg": [cudalike(d["bg_raw"])]}
        if tag == "region":
            batch["alpha_region"] = cudalike(d["alpha_region"])
        pred = B.AnimatingSoftmaxSplatingJoint.forward_flow(me, batch)
        g[f"{tag}_keys"] = np.array(sorted(pred.keys()))
        for k, v in pred.items():
            if tag == "plain" or k not in ("BGImg", "FluidImg"):        # frame-/variant-invariant outputs: stored once
                g[f"{tag}_{k}"] = _plain(v)
    np.savez_compressed(os.path.join(out_dir, "pipeline_v1_surface.npz"), **g)
