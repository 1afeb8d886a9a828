"""The fp32 rung of the decoder convolutions (csrc/conv.hip, SLR_CONV_F32: v_mfma_f32_32x32x2_f32 -- fp32 operands, fp32 products,
fp32 accumulation, the arithmetic of the reference's convolutions, models/layers/partialconv2d.py:61-74, blocks.py:173-248) through
the same entry points, prologue, epilogue and layouts as the split-f16 rung.  Checked against fp64 convolutions (tolerance: fp32
accumulation of Cin * 9 products), against the split-f16 rung, with activations far outside the split's exact range, and on the
whole networks."""
import copy

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


@pytest.mark.parametrize("cin,cout,h,w,bias", [(16, 64, 9, 33, False), (32, 128, 19, 45, True), (64, 64, 64, 96, False),
                                               (48, 192, 8, 32, True), (16, 64, 1, 1, True), (3, 32, 17, 40, True),
                                               (128, 3, 24, 70, True), (3, 3, 16, 32, True), (20, 70, 11, 35, False),
                                               (128, 128, 40, 64, True), (256, 256, 16, 32, False)])
def test_conv3x3_fp32_rung_vs_fp64(S, cin, cout, h, w, bias):
    """slr_conv3x3_forward with SLR_CONV_F32 vs an fp64 convolution: the cases of the split-f16 test (ragged sizes, padded channel
    counts, the 128- / 64- / 32-channel workgroup variants, bias, BN + ReLU prologue, residual, batch of 2) + 128 -> 128 and 256 -> 256."""
    from slr_sfs_amd import nets
    torch.manual_seed(cin + h)
    conv = nets.Conv(cin, cout, 3, bias=bias).cuda()
    if bias:
        conv.bias.data.normal_()
    x = torch.randn(2, cin, h, w, device="cuda") * 3
    sc, sh = torch.rand(cin, device="cuda") + 0.5, torch.randn(cin, device="cuda")
    with torch.no_grad(), nets.fp32_kernels(winograd=False):
        y = conv(x)
        ref = F.conv2d(x.double(), conv.weight.double(), conv.bias.double() if bias else None, padding=1)
        yb = conv(x, (sc, sh))
        xb = F.relu(x * sc.view(1, -1, 1, 1) - sh.view(1, -1, 1, 1))          # fp32, as the prologue computes it
        refb = F.conv2d(xb.double(), conv.weight.double(), conv.bias.double() if bias else None, padding=1)
        res = torch.randn_like(y)
        yr, refr = conv(x, None, res), ref + res.double()
    assert conv.__dict__.get("_wf32") is not None and conv.__dict__.get("_wsplit") is None      # the fp32 rung ran, nothing else
    for got, want in ((y, ref), (yb, refb), (yr, refr)):
        assert (got - want).abs().max().item() < 4e-6 * max(want.abs().max().item(), 1.0)
    # the same layer on the split-f16 rung agrees to the split's accuracy
    with torch.no_grad():
        ys = conv(x)
    assert (ys - y).abs().max().item() < 8e-6 * max(ref.abs().max().item(), 1.0)


@pytest.mark.parametrize("cin,cout,h,w,bias", [(16, 64, 9, 33, False), (32, 128, 19, 45, True), (64, 64, 64, 96, False),
                                               (48, 192, 8, 32, True), (16, 64, 1, 1, True), (3, 32, 17, 40, True),
                                               (20, 70, 11, 35, False), (128, 128, 40, 64, True), (256, 256, 16, 32, False),
                                               (65, 8, 13, 31, True)])
def test_conv3x3_winograd_fp32_vs_fp64(S, cin, cout, h, w, bias):
    """slr_conv3x3_forward with SLR_CONV_F32 | SLR_CONV_WINO (Winograd F(2x2, 3x3) on the fp32 matrix instructions, csrc/conv_wino.hpp)
    vs an fp64 convolution: ragged sizes (odd widths and heights: tiles cut by the image edge), padded channel counts, bias, BN + ReLU
    prologue, residual, batch of 2, NCHW and channel-blocked activations.  Tolerance: 2e-5 of the output range (the transform domain
    amplifies fp32 rounding by a small factor: the direct fp32 rung meets 4e-6 on the same cases); and against the direct rung itself."""
    from slr_sfs_amd import nets
    torch.manual_seed(cin + h)
    conv = nets.Conv(cin, cout, 3, bias=bias).cuda()
    if bias:
        conv.bias.data.normal_()
    x = torch.randn(2, cin, h, w, device="cuda") * 3
    sc, sh = torch.rand(cin, device="cuda") + 0.5, torch.randn(cin, device="cuda")
    with torch.no_grad(), nets.fp32_kernels(winograd=True):
        y = conv(x)
        ref = F.conv2d(x.double(), conv.weight.double(), conv.bias.double() if bias else None, padding=1)
        yb = conv(x, (sc, sh))
        xb = F.relu(x * sc.view(1, -1, 1, 1) - sh.view(1, -1, 1, 1))
        refb = F.conv2d(xb.double(), conv.weight.double(), conv.bias.double() if bias else None, padding=1)
        res = torch.randn_like(y)
        yr, refr = conv(x, None, res), ref + res.double()
    assert conv.__dict__.get("_wwino") is not None and conv.__dict__.get("_wf32") is None and conv.__dict__.get("_wsplit") is None
    errs = []
    for got, want in ((y, ref), (yb, refb), (yr, refr)):
        errs.append((got - want).abs().max().item() / max(want.abs().max().item(), 1.0))
        assert errs[-1] < 2e-5, errs
    with torch.no_grad(), nets.fp32_kernels(winograd=False):
        yd = conv(x)
    assert (yd - y).abs().max().item() < 2e-5 * max(ref.abs().max().item(), 1.0)
    print(f"winograd {cin}->{cout} {h}x{w}: max err / range vs fp64 {max(errs):.2e}")


@pytest.mark.parametrize("cin,cout,h,w,bias", [(64, 128, 16, 40, False), (128, 256, 9, 33, True), (3, 32, 7, 19, True),
                                               (256, 128, 8, 16, False), (64, 65, 5, 27, True), (20, 300, 6, 10, True)])
def test_conv1x1_fp32_rung_vs_fp64(S, cin, cout, h, w, bias):
    from slr_sfs_amd import nets
    torch.manual_seed(cin + cout)
    conv = nets.Conv(cin, cout, 1, bias=bias).cuda()
    if bias:
        conv.bias.data.normal_()
    x = torch.randn(2, cin, h, w, device="cuda") * 2
    with torch.no_grad(), nets.fp32_kernels():
        y = conv(x)
        ref = F.conv2d(x.double(), conv.weight.double(), conv.bias.double() if bias else None)
    assert conv.__dict__.get("_wf32") is not None
    assert (y - ref).abs().max().item() < 4e-6 * max(ref.abs().max().item(), 1.0)


@pytest.mark.parametrize("wino", [False, True])
@pytest.mark.parametrize("cin,cout,h,w,mode", [(64, 64, 24, 70, "derived"), (65, 128, 12, 40, "derived"), (64, 128, 16, 64, "plane"),
                                               (128, 3, 9, 40, "plane"), (32, 32, 20, 33, "chain")])
def test_pconv3x3_fp32_rung_fused_equals_staged(S, cin, cout, h, w, mode, wino):
    """The one-kernel partial convolution on the fp32 rung against the staged path on the same rung (slr_bn_relu_mask ->
    slr_conv3x3_forward -> slr_pconv_epilogue): same operations in the same order on the same accumulators -- bit-exact, update
    mask included, with residual and with next-BN fusion."""
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
    with torch.no_grad(), nets.fp32_kernels(winograd=wino):      # (both 3x3 kernels of the rung: direct, Winograd)
        for kw in ({"residual": res}, {"next_bn": (nsc, nsh)}, {}):
            out, um = pc(x, mask, pre_bn=pre, **kw)
            xin = nets.bn_relu_mask(x, sc, sh, mask) if pre is not None else x
            mplane, mscale = ((x != 0).sum(1, keepdim=True).float(), 1.0) if mask is None else (mask, float(cin))
            box = F.avg_pool2d(mplane, 3, stride=1, padding=1, divisor_override=1)
            raw0 = nets.Conv.conv(pc, xin, None)
            out2, um2 = nets.pconv_epilogue(raw0, pc.bias, box, mscale, cin * 9, kw.get("residual"), kw.get("next_bn"))
            assert torch.equal(um, um2)
            assert torch.equal(out, out2), (out - out2).abs().max().item()


def test_fp32_rung_has_no_magnitude_limit_and_needs_unit_scales(S):
    """Activations of 1e6 -- far outside the split's exact range at any activation scale -- are multiplied exactly like small ones
    (relative error of fp32 accumulation, the saturation counter does not move); the operand scales of the split rung are refused."""
    import ctypes
    from slr_sfs_amd import nets, _lib
    torch.manual_seed(5)
    conv = nets.Conv(64, 128, 3).cuda()
    x = torch.randn(1, 64, 24, 40, device="cuda") * 1.0e6
    nets.saturation_count(x.device)
    with torch.no_grad(), nets.fp32_kernels(winograd=False):
        y = conv(x)
    ref = F.conv2d(x.double(), conv.weight.double(), conv.bias.double(), padding=1)
    assert nets.saturation_count(x.device) == 0
    assert (y - ref).abs().max().item() < 4e-6 * ref.abs().max().item()
    L = _lib.lib()
    with nets.fp32_kernels(winograd=False):
        buf = conv._split_weights()[0]
    out = torch.empty_like(y)
    rc = L.slr_conv3x3_forward(_lib.ptr(x), _lib.ptr(buf), None, None, _lib.ptr(out), 1, 64, 128, 24, 40, 2.0, 64.0, None, None,
                               nets.CONV_F32, _lib.stream_of(x))
    assert rc == -1 and b"fp32 rung" in L.slr_last_error()


def test_decoder_on_the_fp32_rung_vs_fp64(S):
    """The partial-conv decoder with every convolution on the fp32 rung vs its torch definition in fp64: the error of fp32 arithmetic
    itself (the same class as torch's fp32 evaluation on the CPU), on inputs with holes."""
    from slr_sfs_amd import nets
    torch.manual_seed(3)
    dec = nets.DecoderPconv2(64, 3).eval()
    with torch.no_grad():
        for m in dec.modules():
            if hasattr(m, "stored_mean"):
                m.stored_mean.normal_(0, 0.3)
                m.stored_var.uniform_(0.5, 1.5)
        x = torch.randn(1, 64, 72, 136)
        x[:, :, 20:50, 30:80] = 0
        with nets.cpu_reference():
            y32 = dec(x)
            y64 = copy.deepcopy(dec).double()(x.double())
        with nets.fp32_kernels():
            y = dec.cuda()(x.cuda()).cpu()
    e_hip, e_f32 = (y.double() - y64).abs().max().item(), (y32.double() - y64).abs().max().item()
    assert e_hip < 3e-5 and e_hip < 6 * e_f32 + 1e-6, (e_hip, e_f32)


@pytest.mark.parametrize("policy", ["fp32", "fp32-winograd"])
def test_animator_policy_fp32_uses_own_kernels(S, policy):
    """convs="fp32" / "fp32-winograd" render a clip on the fp32 rung without entering torch's convolutions: F.conv2d is never called,
    the frames agree with the split-f16 policy to the split's accuracy."""
    from slr_sfs_amd import pipeline
    torch.manual_seed(0)
    H, W, N = 64, 96, 6
    img = torch.rand(1, 3, H, W, device="cuda") * 2 - 1
    y, x = torch.meshgrid(torch.arange(H, device="cuda", dtype=torch.float32), torch.arange(W, device="cuda", dtype=torch.float32), indexing="ij")
    motion = torch.stack([1.2 * torch.sin(x / 11 + y / 17), 0.9 * torch.cos(x / 13 - y / 7)])[None].contiguous()
    an = pipeline.BaselineAnimator().cuda().eval()
    calls = []
    orig = F.conv2d
    F.conv2d = lambda *a, **k: (calls.append(1), orig(*a, **k))[1]
    try:
        f32 = an.synthesize(img, motion, N, convs=policy)
    finally:
        F.conv2d = orig
    assert not calls
    split = an.synthesize(img, motion, N, convs="split")
    assert float((f32 - split).abs().max()) < 1e-4
