"""The 3x3 convolution onto at most 4 output channels (csrc/conv_few.hpp: fp32 FMAs on the vector ALUs, taken by BOTH rungs -- the 128 -> 3
end of the reference's decoders, models/networks/architectures.py:345-375, blocks.py:173-248): against fp64 convolutions on ragged sizes, every
output-channel count, NCHW and channel-blocked inputs, with prologue / bias / residual; the fused partial convolution bit-exact against the
staged kernels; identical results on the two rungs."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def S():
    import slr_sfs_amd
    slr_sfs_amd._lib.lib()
    return slr_sfs_amd


@pytest.mark.parametrize("cin,cout,h,w", [(3, 1, 17, 70), (20, 2, 33, 130), (64, 3, 16, 64), (128, 3, 40, 200), (24, 4, 19, 67), (8, 3, 1, 1),
                                          (130, 3, 18, 66)])
def test_few_channel_conv_vs_fp64_and_rungs_agree(S, cin, cout, h, w):
    from slr_sfs_amd import nets
    torch.manual_seed(cin * 7 + cout)
    conv = nets.Conv(cin, cout, 3, bias=True).cuda()
    conv.bias.data.normal_()
    x = torch.randn(2, cin, h, w, device="cuda") * 3
    sc, sh = torch.rand(cin, device="cuda") + 0.5, torch.randn(cin, device="cuda")
    res = torch.randn(2, cout, h, w, device="cuda")
    with torch.no_grad():
        ref = F.conv2d(x.double(), conv.weight.double(), conv.bias.double(), padding=1)
        xb = F.relu(x * sc.view(1, -1, 1, 1) - sh.view(1, -1, 1, 1))
        refb = F.conv2d(xb.double(), conv.weight.double(), conv.bias.double(), padding=1)
        outs = []
        for ctx in (torch.no_grad, nets.fp32_kernels):
            with ctx():
                outs.append((conv(x), conv(x, (sc, sh)), conv(x, None, res)))
    for y, yb, yr in outs:
        for got, want in ((y, ref), (yb, refb), (yr, ref + res.double())):
            assert (got - want).abs().max().item() < 2e-6 * max(want.abs().max().item(), 1.0)
    for a, b in zip(*outs):                                     # the same kernel on both rungs
        assert torch.equal(a, b)
    # activations far outside the split rung's exact range: plain fp32, nothing saturates
    nets.saturation_count(x.device)
    with torch.no_grad():
        big = conv(x * 1.0e5)
    assert nets.saturation_count(x.device) == 0
    assert (big - F.conv2d((x * 1.0e5).double(), conv.weight.double(), conv.bias.double(), padding=1)).abs().max().item() < 2e-6 * big.abs().max().item()


@pytest.mark.parametrize("cin,cout,h,w,mode", [(64, 3, 24, 70, "derived"), (65, 1, 12, 40, "derived"), (128, 3, 17, 130, "plane"),
                                               (64, 4, 16, 64, "plane"), (32, 2, 20, 33, "chain")])
def test_few_channel_pconv_fused_equals_staged(S, cin, cout, h, w, mode):
    """The one-kernel partial convolution (prologue + convolution + partial-conv epilogue + mask update, NCHW and channel-blocked inputs)
    against the staged path slr_bn_relu_mask -> slr_conv3x3_forward -> slr_pconv_epilogue: bit-exact, with residual and with next-BN."""
    from slr_sfs_amd import nets
    torch.manual_seed(cout + w)
    pc = nets.PartialConv(cin, cout, 3).cuda()
    pc.bias.data.normal_()
    x = torch.randn(2, cin, h, w, device="cuda")
    x[:, :, 3:8, 5:20] = 0
    sc, sh = torch.rand(cin, device="cuda") + 0.5, torch.randn(cin, device="cuda") * 0.3
    nsc, nsh = torch.rand(cout, device="cuda") + 0.5, torch.randn(cout, device="cuda") * 0.3
    res = torch.randn(2, cout, h, w, device="cuda")
    mask = None if mode == "derived" else (torch.rand(2, 1, h, w, device="cuda") > 0.3).float()
    pre = None if mode == "chain" else (sc, sh)
    with torch.no_grad():
        for kw in ({"residual": res}, {"next_bn": (nsc, nsh)}, {}):
            out, um = pc(x, mask, pre_bn=pre, **kw)
            xin = nets.bn_relu_mask(x, sc, sh, mask) if pre is not None else x
            mplane, mscale = ((x != 0).sum(1, keepdim=True).float(), 1.0) if mask is None else (mask, float(cin))
            box = F.avg_pool2d(mplane, 3, stride=1, padding=1, divisor_override=1)
            raw0 = nets.Conv.conv(pc, xin, None)
            out2, um2 = nets.pconv_epilogue(raw0, pc.bias, box, mscale, cin * 9, kw.get("residual"), kw.get("next_bn"))
            assert torch.equal(um, um2)
            assert torch.equal(out, out2), (out - out2).abs().max().item()
            if mask is not None and cin % 8 == 0:                # the channel-blocked input the decoder hands this layer
                xb8 = x.view(2, cin // 8, 8, h, w).permute(0, 1, 3, 4, 2).contiguous().view(2, cin, h, w)
                out3, um3 = pc(xb8, mask, pre_bn=pre, layout=nets.IN_B8, **kw)
                assert torch.equal(um3, um) and torch.equal(out3, out)
