#!/usr/bin/env python
"""tests/golden/large_nets_e2e.npz: digests of the reference's own network classes and of the FRAMES of its own models at
256x256 -- a grid on which this build's convolution kernels run their multi-tile / channel-blocked 128-channel variants
(conv3x3_split_kernel<1,4,*,true>: two thirds of a timed clip), which the 16x24 / 32x48 fixtures of
tests/golden/nets_reference.npz and the 64x64 frames of pipeline_e2e.npz never reach.

Same recipe as tools/make_golden_nets.py / make_golden_e2e.py (reference classes built by the reference's option parser,
eval mode, bn_noise_misc forced; deterministic state dicts and inputs from tests/nets_fixture.py; the reference's own
euler_integration, softsplat kernel text and forward_flow for the frames).  Stored per tensor: 4096 seeded positions with
their values, per-plane sums (float64) and the max-abs -- numbers only.  Needs /root/reference (build container only)."""
import os
import sys
import types

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import nets_fixture as NF  # noqa: E402
import make_golden as MG  # noqa: E402

S = 256
NPOS = 4096


def digest(g, tag, x):
    x = np.ascontiguousarray(x, dtype=np.float64 if MG.FP64 else np.float32)
    pos = NF.digest_positions(tag, x.size, NPOS)
    g[f"{tag}_shape"] = np.array(x.shape, np.int64)
    g[f"{tag}_val"] = x.ravel()[pos].astype(np.float32)
    g[f"{tag}_plane_sums"] = x.reshape(-1, x.shape[-2] * x.shape[-1]).astype(np.float64).sum(1)
    g[f"{tag}_absmax"] = np.float32(np.abs(x).max())
    print(tag, x.shape, "absmax", float(np.abs(x).max()))


def main(S=S, N=8, ts=None, v1_ts=None, nets_too=True, out_name="large_nets_e2e.npz"):
    """S x S grid, N-frame clip; ts / v1_ts: the frames whose digests are stored (baseline / 2-layer model)."""
    ss, eim = MG.load_reference()

    def stub(name, **kw):
        m = types.ModuleType(name)
        m.__dict__.update(kw)
        sys.modules.setdefault(name, m)
        return sys.modules[name]
    for n in ("cv2", "av", "lz4framed"):
        stub(n)
    tv = stub("torchvision")
    tv.transforms = stub("torchvision.transforms")
    tv.models = stub("torchvision.models", vgg19=None)
    tv.utils = stub("torchvision.utils")
    torch.Tensor.cuda = lambda self, *a, **k: self
    import models.animating_softmax_splating as A
    import models.animating_softmax_splating_2layers_alpha_seperate as B
    from models.networks import utilities as U
    from options.train_options import ArgumentParser

    base = ("--model_type softmax_splating --refine_model_type resnet_256W8UpDown64_de_resnet_pconv2_nonorm "
            "--pconv pconv_pbn_woresbias --norm_G sync:spectral_batch --train_Z --use_softmax_splatter "
            "--losses 1.0_l1 --W %d" % S)
    v1 = base.replace("softmax_splating ", "softmax_splating_2layers_alpha_seperate ") + \
        (" --bg_refine_model_type resnet_256W8UpDown64BG_nonorm "
         "--alpha_refine_model_type resnet_256W8UpDown64Layers_de_resnet_pconv2_nonorm "
         "--out_channel 65 --ngf 64 --train_bg --train_alpha --use_alpha0_as_blending_weight")
    opt_base, _ = ArgumentParser().parse(base)
    opt_v1, _ = ArgumentParser().parse(v1)
    opt_base.bn_noise_misc = opt_v1.bn_noise_misc = True

    def build(fn, name, opt):
        net = fn(opt).eval()
        ref_sd = net.state_dict()
        keys = list(ref_sd.keys())
        shapes = np.full((len(keys), 4), -1, np.int64)
        for i, k in enumerate(keys):
            shapes[i, :ref_sd[k].dim()] = list(ref_sd[k].shape)
        sd = NF.state_dict(name, keys, shapes)
        net.load_state_dict({k: sd[k].to(ref_sd[k].dtype).reshape(ref_sd[k].shape) for k in keys})
        return net

    g = {"S": np.int32(S), "npos": np.int32(NPOS)}
    cl = MG.cudalike
    # fp64 run: the motion field and euler_integration stay fp32 -- the reference's integration is fp32 by definition (its `.float()` state
    # and round-half-even lookups, euler_integration_manipulator.py:27-38); the displacement maps are widened exactly before the splat
    cl_motion = (lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).clone().as_subclass(MG.CudaLike)) if MG.FP64 else cl
    splatter = ss.ModuleSoftsplat("summation")
    if MG.FP64:
        class Splat64(torch.nn.Module):
            def forward(self, tenInput, tenFlow, tenMetric):
                return ss.FunctionSoftsplat(tenInput, tenFlow.double().contiguous().as_subclass(MG.CudaLike),
                                            None if tenMetric is None else tenMetric.double().as_subclass(MG.CudaLike), "summation")
        splatter = Splat64()
        # the reference's hard-coded fp32 casts (`mask = (x != 0).float()`, architectures.py:369; `float_x = x.float()` in its batch
        # norms, normalization.py:238,321; the alpha scalars, animating_softmax_splating.py:585) widened as well -- everywhere except
        # inside euler_integration, which runs with the real Tensor.float
        real_float = torch.Tensor.float
        widen = lambda self, *a, **k: self.double()
        torch.Tensor.float = widen

        def euler32(*a, **k):
            torch.Tensor.float = real_float
            try:
                return eim.euler_integration(*a, **k)
            finally:
                torch.Tensor.float = widen
        A.euler_integration = B.euler_integration = euler32
    npdt = np.float64 if MG.FP64 else np.float32
    plain = lambda t: t.detach().as_subclass(torch.Tensor).numpy().astype(npdt)
    tin = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=npdt))
    with torch.no_grad():
        # ---- the network classes by themselves
        nets = {"encoder": (U.get_encoder, opt_v1), "projector": (U.get_decoder, opt_v1),
                "net_alpha_decoder": (U.get_alpha_decoder, opt_v1)}
        built = {}
        if not nets_too:
            built["net_alpha_decoder"] = build(U.get_alpha_decoder, "net_alpha_decoder", opt_v1)
        for name, (fn, opt) in (nets.items() if nets_too else ()):
            net = built[name] = build(fn, name, opt)
            out = net(NF.net_input_large(name, S))
            out = out if isinstance(out, tuple) else (out,)
            g[f"{name}_nout"] = np.int32(len(out))
            for i, o in enumerate(out):
                digest(g, f"{name}_out{i}", o.numpy())
        # ---- frames of both models (the runner flow of make_golden_e2e.py) at S x S
        img, motion, N = NF.e2e_inputs(S, N)
        g["N"] = np.int32(N)
        enc, dec = build(U.get_encoder, "encoder", opt_base), build(U.get_decoder, "projector", opt_base)
        fs, Z = enc(tin(img))
        me = types.SimpleNamespace(opt=opt_base, softsplater=splatter, projector=dec)
        ts = [1, N // 2, N - 1] if ts is None else list(ts)
        v1_ts = [N // 2] if v1_ts is None else list(v1_ts)
        g["ts"] = np.array(ts, np.int32)
        g["v1_ts"] = np.array(v1_ts, np.int32)
        for t in ts:
            batch = {"features": [(cl(plain(fs)), cl(plain(Z)))], "images": [cl(img)], "motions": [cl_motion(motion)],
                     "index": torch.tensor([[0, t, N - 1]])}
            digest(g, f"baseline_PredImg_t{t}", plain(A.AnimatingSoftmaxSplating.forward_flow(me, batch)["PredImg"]))
        bgn, aenc, adec = (build(U.get_net_bg, "net_bg", opt_v1), build(U.get_alpha_encoder, "net_alpha_encoder", opt_v1),
                           built["net_alpha_decoder"])
        enc1, dec1 = build(U.get_encoder, "encoder", opt_v1), build(U.get_decoder, "projector", opt_v1)
        fs1, Z1 = enc1(tin(img))
        bg = bgn(tin(img))
        me = types.SimpleNamespace(opt=opt_v1, softsplater=splatter, projector=dec1,
                                   net_alpha_decoder=adec, net_alpha_encoder=aenc)
        for t in v1_ts:
            batch = {"features": [(cl(plain(fs1)), cl(plain(Z1)))], "images": [cl(img)], "motions": [cl_motion(motion)],
                     "index": torch.tensor([[0, t, N - 1]]), "BGImg": [cl(plain(bg))]}
            pred = B.AnimatingSoftmaxSplatingJoint.forward_flow(me, batch)
            for k in ("PredImg", "FluidImg", "CompositeFluidAlpha"):
                digest(g, f"v1_{k}_t{t}", plain(pred[k]))
    p = os.path.join(ROOT, "tests", "golden", out_name)
    np.savez_compressed(p, **g)
    print("wrote", p, os.path.getsize(p) // 1024, "kB")


if __name__ == "__main__":
    import shutil
    try:
        if "--native" in sys.argv[1:] and "--fp64" in sys.argv[1:]:
            # The single-valued arbiter at the native size: the SAME reference classes, forward_flow, euler_integration and
            # splat kernel text run in float64 throughout (torch default dtype float64; the kernel text compiled with -Dfloat=double) --
            # rounding noise 1e-13 instead of the 2e-4 spread of the two fp32 runs.  Stored beside them as `*_val_fp64` /
            # `*_plane_sums_fp64` (values rounded to fp32 for storage: 6e-8).
            MG.FP64 = True
            torch.set_default_dtype(torch.float64)
            main(S=768, N=60, ts=[1, 30, 59], v1_ts=[1, 30, 59], nets_too=False, out_name="native_frames_768_fp64.npz")
            gd = os.path.join(ROOT, "tests", "golden")
            a, b = dict(np.load(os.path.join(gd, "native_frames_768_fp64.npz"))), dict(np.load(os.path.join(gd, "native_frames_768.npz")))
            for k in list(a):
                if k.endswith("_val") or k.endswith("_plane_sums"):
                    b[k + "_fp64"] = a[k]
            np.savez_compressed(os.path.join(gd, "native_frames_768.npz"), **b)
            os.remove(os.path.join(gd, "native_frames_768_fp64.npz"))
        elif "--native" in sys.argv[1:]:
            # the reference's own working size (test_animating/CLAW/test_v1.sh:19: W = 768, N = 60), frames only -- run TWICE, with
            # torch's oneDNN convolutions and with its plain ones (im2col + sgemm): at this size the reference's OWN fp32 frames differ
            # by up to 2.0e-4 between the two (frame 30 of the baseline model: 4 of 4096 sampled values apart by more than 1e-4), i.e.
            # the reference's frame is only defined up to that spread.  Both runs are stored (`*_val` plain, `*_val_onednn`); the test
            # measures the distance to the interval the two runs span.
            main(S=768, N=60, ts=[1, 30, 59], v1_ts=[1, 30, 59], nets_too=False, out_name="native_frames_768_onednn.npz")
            torch.backends.mkldnn.enabled = False
            main(S=768, N=60, ts=[1, 30, 59], v1_ts=[1, 30, 59], nets_too=False, out_name="native_frames_768.npz")
            gd = os.path.join(ROOT, "tests", "golden")
            a, b = dict(np.load(os.path.join(gd, "native_frames_768_onednn.npz"))), dict(np.load(os.path.join(gd, "native_frames_768.npz")))
            for k in list(b):
                if k.endswith("_val") or k.endswith("_plane_sums"):
                    b[k + "_onednn"] = a[k]
            np.savez_compressed(os.path.join(gd, "native_frames_768.npz"), **b)
            os.remove(os.path.join(gd, "native_frames_768_onednn.npz"))
        else:
            main()
    finally:
        shutil.rmtree(MG.TMP, ignore_errors=True)
