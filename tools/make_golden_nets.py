#!/usr/bin/env python
"""tests/golden/nets_reference.npz: outputs of the REFERENCE's own network classes
(models/networks/architectures.py:121-197,233-260,345-375; models/layers/blocks.py:47-87,173-248;
models/layers/partialconv2d.py:41-81) -- get_encoder / get_decoder / get_net_bg / get_alpha_encoder /
get_alpha_decoder built by the reference's option parser with the canonical flag sets
(train_animating_scripts/*.sh), eval mode, bn_noise_misc forced like the test scripts do
(test_baseline_4eval_rawsize.py:127) -- on the deterministic state dicts and inputs of tests/nets_fixture.py.
Stored: key / shape lists of the state dicts and the output tensors; no weights, nothing of the reference's text.
Needs /root/reference (build container only)."""
import os
import sys
import types

import numpy as np
import torch

REF = "/root/reference"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import nets_fixture as NF  # noqa: E402


def main():
    sys.path.insert(0, REF)

    def stub(name, **kw):
        m = types.ModuleType(name)
        m.__dict__.update(kw)
        sys.modules.setdefault(name, m)
        return sys.modules[name]
    stub("cupy", memoize=lambda for_each_device=False: (lambda f: f), cuda=types.SimpleNamespace(compile_with_cache=None))
    for n in ("cv2", "av", "lz4framed"):
        stub(n)
    tv = stub("torchvision")
    tv.transforms = stub("torchvision.transforms")
    tv.models = stub("torchvision.models", vgg19=None)
    tv.utils = stub("torchvision.utils")
    from models.networks import utilities as U
    from options.train_options import ArgumentParser
    flags = ("--model_type softmax_splating_2layers_alpha_seperate "
             "--refine_model_type resnet_256W8UpDown64_de_resnet_pconv2_nonorm --pconv pconv_pbn_woresbias "
             "--norm_G sync:spectral_batch --train_Z --losses 1.0_l1 --W 32 "
             "--bg_refine_model_type resnet_256W8UpDown64BG_nonorm "
             "--alpha_refine_model_type resnet_256W8UpDown64Layers_de_resnet_pconv2_nonorm "
             "--out_channel 65 --ngf 64 --train_bg --train_alpha --use_alpha0_as_blending_weight")
    opt, _ = ArgumentParser().parse(flags)
    opt.bn_noise_misc = True
    build = {"encoder": U.get_encoder, "projector": U.get_decoder, "net_bg": U.get_net_bg,
             "net_alpha_encoder": U.get_alpha_encoder, "net_alpha_decoder": U.get_alpha_decoder}
    g = {}
    for name, fn in build.items():
        net = fn(opt).eval()
        ref_sd = net.state_dict()
        keys = list(ref_sd.keys())
        shapes = np.full((len(keys), 4), -1, np.int64)
        for i, k in enumerate(keys):
            shapes[i, :ref_sd[k].dim()] = list(ref_sd[k].shape)
        sd = NF.state_dict(name, keys, shapes)
        net.load_state_dict({k: sd[k].to(ref_sd[k].dtype).reshape(ref_sd[k].shape) for k in keys})
        with torch.no_grad():
            out = net(NF.net_input(name))
        out = out if isinstance(out, tuple) else (out,)
        g[f"{name}_keys"] = np.array(keys)
        g[f"{name}_shapes"] = shapes
        g[f"{name}_nout"] = np.int32(len(out))
        for i, o in enumerate(out):
            g[f"{name}_out{i}"] = o.numpy().astype(np.float32)
            print(name, i, tuple(o.shape), "max-abs", float(o.abs().max()), "finite", bool(torch.isfinite(o).all()))
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "nets_reference.npz"), **g)
    print("wrote nets_reference.npz", os.path.getsize(os.path.join(ROOT, "tests", "golden", "nets_reference.npz")) // 1024, "kB")


if __name__ == "__main__":
    main()
