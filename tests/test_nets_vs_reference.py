"""The encoder/decoder definitions in slr_sfs_amd.nets against the reference's own classes,
on CPU.  Runs only where /root/reference exists (the build container); skipped on the GPU box.
The reference networks are constructed from its real option parser with the canonical flag
sets (train_animating_scripts/*.sh), random-initialised, their BN statistics randomised, and
their state dict loaded into ours through load_reference_state_dict."""
import os
import sys
import types

import pytest
import torch

REF = "/root/reference"
pytestmark = pytest.mark.skipif(not os.path.isdir(REF), reason="reference checkout not present")


@pytest.fixture(scope="module")
def ref():
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
    opt.bn_noise_misc = True                       # forced by the test scripts (:127)
    yield types.SimpleNamespace(U=U, opt=opt)
    sys.path.remove(REF)
    for k in [k for k in sys.modules if k == "models" or k.startswith("models.") or k.startswith("options")]:
        del sys.modules[k]


def _randomise(net, seed):
    g = torch.Generator().manual_seed(seed)
    for name, buf in net.named_buffers():
        if name.endswith("stored_mean"):
            buf.copy_(torch.randn(buf.shape, generator=g) * 0.2)
        elif name.endswith("stored_var"):
            buf.copy_(torch.rand(buf.shape, generator=g) + 0.5)
    return net.eval()


@torch.no_grad()
def _check(refnet, mine, x, prefix="", tol=2e-4):
    from slr_sfs_amd import nets
    sd = {prefix + k: v for k, v in refnet.state_dict().items()}
    nets.load_reference_state_dict(mine, sd, prefix)
    with nets.cpu_reference():                      # the torch definition of the fused stages, on the CPU
        a, b = refnet(x), mine.eval()(x)
    if not isinstance(a, tuple):
        a, b = (a,), (b,)
    for ra, rb in zip(a, b):
        scale = ra.abs().max().item() + 1e-6
        assert (ra - rb).abs().max().item() <= tol * scale, ((ra - rb).abs().max().item(), scale)


def test_encoder_with_z(ref):
    from slr_sfs_amd import nets
    torch.manual_seed(0)
    r = _randomise(ref.U.get_encoder(ref.opt), 1)
    _check(r, nets.EncoderWithZ(), torch.rand(1, 3, 24, 40) * 2 - 1, prefix="model.module.encoder.")


def test_decoder_pconv2_with_holes(ref):
    from slr_sfs_amd import nets
    torch.manual_seed(1)
    r = _randomise(ref.U.get_decoder(ref.opt), 2)
    x = torch.randn(1, 64, 32, 48)
    x[:, :, 5:20, 10:30] = 0.0                     # a hole: exercises mask update / ratio / resampling
    x[:, 3, 0, 0] = 0.0                            # single-channel zero: per-channel first mask
    _check(r, nets.DecoderPconv2(64, 3), x, prefix="model.module.projector.")
    _check(r, nets.DecoderPconv2(64, 3), torch.zeros(1, 64, 16, 16))     # everything a hole


def test_bg_decoder_and_alpha_nets(ref):
    from slr_sfs_amd import nets
    torch.manual_seed(2)
    img = torch.rand(1, 3, 32, 32) * 2 - 1
    _check(_randomise(ref.U.get_net_bg(ref.opt), 3), nets.BGDecoder(), img)
    _check(_randomise(ref.U.get_alpha_encoder(ref.opt), 4), nets.Encoder(3, 2), img)
    x = torch.randn(1, 65, 32, 32)
    x[:, :, 8:16, 8:24] = 0.0
    _check(_randomise(ref.U.get_alpha_decoder(ref.opt), 5), nets.DecoderPconv2(65, 1), x)
