"""Encoder / decoder networks either side of the splat path, as INFERENCE modules,
so the full configurations C3/C4 of BASELINE.json can be run and real checkpoints load.
On a device every 3x3 (partial) convolution, with the BN / ReLU / mask / ratio / bias / residual
stages around it, is ONE hand-written matrix-core kernel (csrc/conv.hip, split-f16 implicit GEMM),
the 1x1 skip convolutions run on the same arithmetic and the resampling stages are HIP kernels
(csrc/resample.hip): nothing goes to MIOpen.  On the CPU (validation against the reference's own classes,
tests/test_nets_vs_reference.py) the same modules run the torch composition that defines them.

Own definitions, folded for inference (reference file:line cited per class):
  * spectral norm (legacy hook, models/layers/blocks.py:5-18): eval-mode weight is
    weight_orig / (u^T W v), no power iteration -> folded once at load time;
  * noise-conditioned BN (models/layers/normalization.py:19-90) with the zero noise the test
    scripts force (bn_noise_misc=True, test_animating/test_baseline_4eval_rawsize.py:127):
    gain = 1, bias = 0, stored statistics -> a per-channel scale/shift (:219-231);
  * partial convolution (models/layers/partialconv2d.py:41-81, multi_channel=True): the mask
    update conv(mask, ones[out,in,k,k]) is the same for every output channel and equals
    box_filter(sum_c mask); masks are binary so this is exact integer arithmetic in fp32
    (< 2^24) -> computed on ONE channel instead of a second full-size convolution
    (halves the decoder FLOPs without changing a bit of its result).

``load_reference_state_dict`` maps the reference's checkpoint key scheme onto these modules.
"""
import math

import threading

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import _lib


IN_B8, OUT_B8, RES_B8, CONV_F32, CONV_WINO, SKIP_B8, POOL_OUT, UP_OUT = 1, 2, 4, 8, 16, 32, 64, 128      # include/slr_splat.h: SLR_CONV_IN_B8 / _OUT_B8 / _RES_B8 / SLR_CONV_F32 / _WINO / _SKIP_B8 / _POOL_OUT / _UP_OUT


class _Route(threading.local):
    """What the context managers below switch, per HOST THREAD (two animators on two threads do not see each other's route or
    activation scale).  The saturation counter they react to is one per DEVICE (include/slr_splat.h: slr_conv_saturation_count):
    its attribution to a batch assumes one renderer per device at a time."""
    cpu_reference = False
    torch_convs = False                  # inside torch_convolutions(): every stage through its torch definition (fp32)
    act_scale = 64.0                     # pre-scale of the activations in the split-f16 kernels (include/slr_splat.h: xscale)
    f32_kernels = False                  # inside fp32_kernels(): the convolutions on the fp32 matrix instructions (the fp32 rung)
    winograd = False                     # ... its 3x3 convolutions as Winograd F(2x2, 3x3) (csrc/conv_wino.hpp); False (= FP32_WINOGRAD): direct
    fused_skips = True                   # a block's 1x1 skip convolution rides in its second 3x3 kernel (slr_*_forward_skip); staged_skips(): two kernels
    fused_pools = True                   # ... and a "Down" block's average pool in that kernel's epilogue; staged_skips(pools_only=True): pool kernel


_S = _Route()


def _b8(x, channels):
    """On the device every activation whose channel count is a multiple of 8 lives channel-blocked ([N,C/8,H,W,8]
    in memory, carried in a tensor of the logical shape [N,C,H,W]): all its producers and consumers are our kernels,
    which then move 16 bytes per lane and instruction instead of 4.  The networks' inputs and outputs (3-, 65-, 2-channel
    ends, the splat's feature planes) stay NCHW; no torch op ever touches a blocked tensor."""
    return x.is_cuda and channels % 8 == 0 and not _S.torch_convs


class torch_convolutions:
    """VALIDATION AID, not a product route: no animator policy selects it (pipeline.CONV_POLICIES = auto | split | fp32 | fp32-winograd, all on this
    package's kernels).  Inside it every stage runs the torch composition that DEFINES it (nets.py: F.conv2d -> MIOpen fp32 on ROCm,
    elementwise stages as torch ops) -- the arithmetic the reference's decoder uses (models/layers/partialconv2d.py:61-74,
    models/networks/architectures.py:345-375).  The GPU tests compare the HIP stages with it, and bench.py times it beside the fp32
    rung (fps_fp32_convs.through_torch_miopen)."""

    def __enter__(self):
        self._prev, _S.torch_convs = _S.torch_convs, True
        return self

    def __exit__(self, *exc):
        _S.torch_convs = self._prev
        return False


FP32_WINOGRAD = False                    # what fp32_kernels() without an argument selects (the strict rung: direct 3x3 kernels)


class fp32_kernels:
    """Context manager: the FULL-RANGE fp32 rung of these networks on this package's own kernels.  Inside it every 3x3 / 1x1
    convolution runs on v_mfma_f32_32x32x2_f32 (csrc/conv.hip, SLR_CONV_F32): fp32 operands, fp32 products, fp32 accumulation --
    the arithmetic of the reference's decoder (models/layers/partialconv2d.py:61-74, models/networks/architectures.py:345-375) with
    no limit on the magnitude of the activations, at the fp32 matrix rate (157 TFLOP/s against the ~830 effective of the split-f16
    rung); prologue, epilogue, mask update, layouts and every other kernel are the ones of the split-f16 rung.  The animators enter
    it on request (convs="fp32") or by themselves when the split-f16 kernels report a clamped activation (convs="auto").
    winograd=True (the animators' convs="fp32-winograd"): the 3x3 convolutions with more than 4 output channels as Winograd F(2x2, 3x3) on
    the same instructions (csrc/conv_wino.hpp, SLR_CONV_WINO) -- 16 instead of 36 multiplications per 2x2 outputs, still fp32 operands /
    products / accumulation, 1.5x the direct kernel's speed.  Accuracy, measured: per layer its error is 2 - 4x the direct kernel's
    (<= 1.1e-6 of the output range against <= 3e-7); the seeded random-weight networks of the native fixture amplify a perturbation of a
    layer about 400x at a few ill-conditioned pixels (the reference's own two fp32 runs differ by 2e-4 there), so whole frames come out
    9e-7 from the direct rung's on average and up to 3.9e-4 at ~50 pixels -- 2.6e-4 from the fp64 frames at worst, where the direct
    rung (the default, winograd=False / nets.FP32_WINOGRAD) stays within 1.4e-5 (tests/test_large_golden.py).  Fast fp32, not the anchor."""

    def __init__(self, winograd=None):
        self.winograd = FP32_WINOGRAD if winograd is None else bool(winograd)

    def __enter__(self):
        self._prev, _S.f32_kernels = _S.f32_kernels, True
        self._prev_w, _S.winograd = _S.winograd, self.winograd
        return self

    def __exit__(self, *exc):
        _S.f32_kernels = self._prev
        _S.winograd = self._prev_w
        return False


class activation_scale:
    """Context manager: pre-scale of the activations in the split-f16 kernels (a power of two in (0, 64]; default 64).
    The exact domain of the split is |activation| < 65472 / scale: 1023 at 64, 65472 at 1 (csrc/conv.hip)."""

    def __init__(self, scale):
        m, e = math.frexp(float(scale))
        if not (0.0 < scale <= 64.0 and m == 0.5):
            raise ValueError("activation_scale: a power of two in (0, 64]")
        self.scale = float(scale)

    def __enter__(self):
        self._prev, _S.act_scale = _S.act_scale, self.scale
        return self

    def __exit__(self, *exc):
        _S.act_scale = self._prev
        return False


class staged_skips:
    """VALIDATION AID: inside, a residual block's 1x1 skip convolution is a kernel of its own again (slr_conv1x1_forward, its result the
    residual of the second 3x3 kernel) -- the form the fused kernels are tested against (tests/test_gpu_parity.py)."""

    def __init__(self, pools_only=False):
        self.pools_only = pools_only                 # keep the fused skip, stage only the "Down" blocks' average pool

    def __enter__(self):
        self._prev = (_S.fused_skips, _S.fused_pools)
        _S.fused_skips = self.pools_only
        _S.fused_pools = False

    def __exit__(self, *exc):
        _S.fused_skips, _S.fused_pools = self._prev
        return False


def _skip_rides(a, cout, b8, b8_in):
    """The skip branch can ride in the second 3x3 kernel: split-f16 rung, channel-blocked block input and intermediate, more than 4
    output channels (include/slr_splat.h: slr_conv3x3_forward_skip)."""
    if _S.f32_kernels and (_S.winograd or cout <= 64):         # fp32 rung: the 128-channel kernels only; none in the Winograd kernel
        return False
    return bool(_S.fused_skips and b8 and b8_in and cout > 4 and a.is_cuda and not _S.torch_convs)


def _pool_out(N, cout, H, W, like, pool):
    """(output tensor, pooling scratch, its bytes, layout flag) of the *_forward_skip calls: with ``pool`` the output is the block's
    avgpool3x3s2 result and the kernel needs side buffers (include/slr_splat.h: SLR_CONV_POOL_OUT)."""
    if not pool:
        return torch.empty(N, cout, H, W, device=like.device, dtype=like.dtype), None, 0, 0
    if pool == "Up":                                  # x2 bilinear up-sampling in the epilogue (SLR_CONV_UP_OUT)
        nbytes = _lib.lib().slr_conv_up_ws_bytes(N, cout, H, W)
        ws = torch.empty(nbytes, dtype=torch.uint8, device=like.device)
        return torch.empty(N, cout, 2 * H, 2 * W, device=like.device, dtype=like.dtype), ws, nbytes, UP_OUT
    nbytes = _lib.lib().slr_conv_pool_ws_bytes(N, cout, H, W)
    ws = torch.empty(nbytes, dtype=torch.uint8, device=like.device)
    return torch.empty(N, cout, (H - 1) // 2 + 1, (W - 1) // 2 + 1, device=like.device, dtype=like.dtype), ws, nbytes, POOL_OUT


class cpu_reference:
    """Context manager for the TESTS that validate these module definitions against the reference's own
    classes (which only exist where the reference checkout is, on the CPU): inside it, CPU tensors run
    the torch composition that defines every fused stage.  Outside it CPU tensors raise, like the
    reference's operators do (models/softsplat.py:418-419): the product has no CPU path.  (bench.py's `cpu_baseline` leg
    also enters it, to TIME the decoder on the host cores for its reported CPU figure -- never a product path.)"""

    def __enter__(self):
        self._prev, _S.cpu_reference = _S.cpu_reference, True
        return self

    def __exit__(self, *exc):
        _S.cpu_reference = self._prev
        return False


def _fused_ok(*ts):
    """Device tensors ALWAYS take the HIP kernels (fp32, contiguous, no autograd -- anything else on a
    device raises: there is no silent torch path on the GPU).  CPU tensors raise NotImplementedError
    unless the caller is inside ``cpu_reference()`` (validation of these definitions against the
    reference classes): then they take the torch composition, which IS the definition the kernels are
    tested against (tests/test_gpu_parity.py)."""
    if _S.torch_convs and all(t.is_cuda for t in ts):
        return False                                       # the supported fp32 route (torch_convolutions)
    if not any(t.is_cuda for t in ts):
        if not _S.cpu_reference:
            raise NotImplementedError("slr_sfs_amd.nets run on ROCm device tensors only (no CPU path); the torch "
                                      "definition used for validation is available inside nets.cpu_reference()")
        return False
    if not all(t.is_cuda and t.dtype == torch.float32 and t.is_contiguous() for t in ts):
        raise ValueError("slr_sfs_amd.nets: fused decoder stages need fp32 contiguous tensors on one device")
    if torch.is_grad_enabled() and any(t.requires_grad for t in ts):
        raise RuntimeError("slr_sfs_amd.nets are inference modules: run them under torch.no_grad()")
    return True


def bn_relu_mask(x, scale, shift, mask):
    """relu(x*scale - shift) * mask;  mask None = (x != 0), mask False = no mask."""
    if _fused_ok(x, *([mask] if torch.is_tensor(mask) else [])):
        N, C, H, W = x.shape
        y = torch.empty_like(x)
        mc = -1 if mask is False else 0 if mask is None else mask.shape[1]
        with torch.cuda.device(x.device):
            _lib.check(_lib.lib().slr_bn_relu_mask(_lib.ptr(x), _lib.ptr(scale), _lib.ptr(shift),
                                                   _lib.ptr(mask) if torch.is_tensor(mask) else None, mc, _lib.ptr(y),
                                                   N, C, H, W, _lib.stream_of(x)), "slr_bn_relu_mask")
        return y
    y = F.relu(x * scale.view(1, -1, 1, 1) - shift.view(1, -1, 1, 1))
    if mask is False:
        return y
    return y * ((x != 0).to(x.dtype) if mask is None else mask)


def pconv_epilogue(raw0, bias, mask_box, mask_scale, winsize, residual=None, next_bn=None):
    """(raw0*ratio + b)*um on the bias-free convolution output, then `+ residual` or the next
    convolution's relu(bn(.))*um  (partialconv2d.py:61-74, blocks.py:233-236,248).
    um_raw = mask_box*mask_scale.  Returns (out, um)."""
    if _fused_ok(raw0, mask_box, *([] if residual is None else [residual])):
        N, C, H, W = raw0.shape
        out = torch.empty_like(raw0)
        um = torch.empty_like(mask_box)
        sc, sh = next_bn if next_bn is not None else (None, None)
        with torch.cuda.device(raw0.device):
            _lib.check(_lib.lib().slr_pconv_epilogue(_lib.ptr(raw0), _lib.ptr(bias), _lib.ptr(mask_box), float(mask_scale),
                                                     _lib.ptr(residual), _lib.ptr(sc), _lib.ptr(sh), _lib.ptr(out),
                                                     _lib.ptr(um), float(winsize), N, C, H, W,
                                                     _lib.stream_of(raw0)), "slr_pconv_epilogue")
        return out, um
    um_raw = mask_box * mask_scale
    um = torch.clamp(um_raw, 0, 1)
    ratio = winsize / (um_raw + 1e-8) * um
    out = (raw0 * ratio + bias.view(1, -1, 1, 1)) * um
    if residual is not None:
        out = out + residual
    if next_bn is not None:
        out = F.relu(out * next_bn[0].view(1, -1, 1, 1) - next_bn[1].view(1, -1, 1, 1)) * um
    return out, um


# --------------------------------------------------------------------------- building blocks

class AffineBN(nn.Module):
    """Eval-mode noise-BN with zero noise: y = x*scale - shift, scale = rsqrt(var+eps),
    shift = mean*scale (fused_bn, models/layers/normalization.py:219-231)."""

    def __init__(self, ch, eps=1e-5):
        super().__init__()
        self.eps = eps
        self.register_buffer("stored_mean", torch.zeros(ch))
        self.register_buffer("stored_var", torch.ones(ch))

    def scale_shift(self):
        """(scale, shift), computed once per device (statistics are frozen at inference; loading a
        state dict resets the cache)."""
        c = self.__dict__.get("_ss")
        if c is None or c[0].device != self.stored_var.device:
            scale = torch.rsqrt(self.stored_var + self.eps)
            c = self.__dict__["_ss"] = (scale, self.stored_mean * scale)
        return c

    def forward(self, x):
        scale, shift = self.scale_shift()
        return x * scale.view(1, -1, 1, 1) - shift.view(1, -1, 1, 1)


class Conv(nn.Module):
    """Convolution with its spectral normalisation already folded into ``weight``."""

    def __init__(self, cin, cout, k, bias=True):
        super().__init__()
        self.weight = nn.Parameter(torch.empty(cout, cin, k, k), requires_grad=False)
        self.bias = nn.Parameter(torch.zeros(cout), requires_grad=False) if bias else None
        self.pad = k // 2
        self.cin, self.k = cin, k
        nn.init.normal_(self.weight, std=math.sqrt(1.0 / (cin * k * k)))

    def forward(self, x, pre_bn=None, residual=None, layout=0):
        return self.conv(x, self.bias, pre_bn, residual, layout)

    def _w4(self):
        """This 1x1 convolution's weights as plain fp32 [Cin][4] (row ci = w[0..3][ci], zero padded): the skip operand of the <= 4-channel
        3x3 kernel (slr_*_forward_skipout); prepared once per device / weight version."""
        w = self.weight
        key = (w.data_ptr(), w._version, w.device)
        c = self.__dict__.get("_w4c")
        if c is None or c[0] != key:
            w4 = torch.zeros(w.shape[1], 4, device=w.device, dtype=torch.float32)
            w4[:, :w.shape[0]] = w.view(w.shape[0], w.shape[1]).t()
            c = self.__dict__["_w4c"] = (key, w4.contiguous())
        return c[1]

    def forward_skipout(self, x, pre_bn, skip_conv, layout=0):
        """(conv(relu(bn(x))) + bias, skip_conv(x)) from ONE pass over x: Cout <= 4, channel-blocked x (slr_conv3x3_forward_skipout)."""
        assert self.k == 3 and skip_conv.k == 1 and _fused_ok(x)
        cout, cin = self.weight.shape[:2]
        N, _, H, W = x.shape
        buf, wscale, xscale, arith = self._split_weights()
        out = torch.empty(N, cout, H, W, device=x.device, dtype=x.dtype)
        skip = torch.empty(N, cout, H, W, device=x.device, dtype=x.dtype)
        sc, sh = pre_bn if pre_bn is not None else (None, None)
        with torch.cuda.device(x.device):
            _lib.check(_lib.lib().slr_conv3x3_forward_skipout(
                _lib.ptr(x), _lib.ptr(buf), _lib.ptr(self.bias), None, _lib.ptr(out), N, cin, cout, H, W, wscale, xscale, _lib.ptr(sc), _lib.ptr(sh),
                _lib.ptr(skip_conv._w4()), _lib.ptr(skip_conv.bias), _lib.ptr(skip), layout | arith, _lib.stream_of(x)), "slr_conv3x3_forward_skipout")
        return out, skip

    def forward_skip(self, x, pre_bn, skip_x, skip_conv, layout=0, skip_b8=False, pool=False):
        """conv(relu(bn(x))) + bias + skip_conv(skip_x) in ONE kernel (slr_conv3x3_forward_skip; blocks.py:83-87); ``pool``: and the
        "Down" block's average pool (:196-199) in its epilogue."""
        assert self.k == 3 and skip_conv.k == 1 and _fused_ok(x, skip_x)
        cout, cin = self.weight.shape[:2]
        N, _, H, W = x.shape
        buf, wscale, xscale, arith = self._split_weights()
        sbuf, swscale, _, sarith = skip_conv._split_weights()
        assert arith == sarith and not (arith & CONV_WINO)
        out, ws, ws_bytes, pflag = _pool_out(N, cout, H, W, x, pool)
        sc, sh = pre_bn if pre_bn is not None else (None, None)
        with torch.cuda.device(x.device):
            _lib.check(_lib.lib().slr_conv3x3_forward_skip(
                _lib.ptr(x), _lib.ptr(buf), _lib.ptr(self.bias), _lib.ptr(out), N, cin, cout, H, W, wscale, xscale, _lib.ptr(sc), _lib.ptr(sh),
                _lib.ptr(skip_x), _lib.ptr(sbuf), _lib.ptr(skip_conv.bias), skip_x.shape[1], swscale, _lib.ptr(ws), ws_bytes,
                layout | arith | pflag | (SKIP_B8 if skip_b8 else 0), _lib.stream_of(x)), "slr_conv3x3_forward_skip")
        return out

    def _split_weights(self):
        """Weights of the matrix-core kernels (csrc/conv.hip) in fragment order, prepared once per device / weight version and
        arithmetic: (buffer, wscale, xscale, layout flag).  Split-f16 rung: hi / lo halves scaled by wscale; fp32 rung
        (inside fp32_kernels()): the fp32 values themselves, no scales."""
        w = self.weight
        f32 = _S.f32_kernels
        wino = f32 and _S.winograd and self.k == 3 and w.shape[0] > 4 and w.shape[1] <= 256     # (its prologue table holds 256 channels)
        key = (w.data_ptr(), w._version, w.device)
        name = "_wwino" if wino else "_wf32" if f32 else "_wsplit"
        c = self.__dict__.get(name)
        if c is None or c[0] != key:
            L = _lib.lib()
            nbytes = L.slr_conv3x3_wino_weight_bytes if wino else L.slr_conv3x3_weight_bytes if self.k == 3 else L.slr_conv1x1_weight_bytes
            buf = torch.empty(nbytes(w.shape[0], w.shape[1]), dtype=torch.uint8, device=w.device)
            with torch.cuda.device(w.device):
                if wino:
                    wscale = 1.0
                    _lib.check(L.slr_conv3x3_wino_weights(_lib.ptr(w), _lib.ptr(buf), w.shape[0], w.shape[1], _lib.stream_of(w)), "slr_conv3x3_wino_weights")
                elif f32:
                    prep = L.slr_conv3x3_f32_weights if self.k == 3 else L.slr_conv1x1_f32_weights
                    wscale = 1.0
                    _lib.check(prep(_lib.ptr(w), _lib.ptr(buf), w.shape[0], w.shape[1], _lib.stream_of(w)), "slr_conv_f32_weights")
                else:
                    amax = float(w.abs().max())
                    wscale = 2.0 ** math.floor(math.log2(4096.0 / amax)) if amax > 0 else 1.0
                    split = L.slr_conv3x3_split_weights if self.k == 3 else L.slr_conv1x1_split_weights
                    _lib.check(split(_lib.ptr(w), _lib.ptr(buf), w.shape[0], w.shape[1], wscale, _lib.stream_of(w)),
                               "slr_conv_split_weights")
            c = self.__dict__[name] = (key, buf, wscale)
        return (c[1], 1.0, 1.0, CONV_F32 | (CONV_WINO if wino else 0)) if f32 else (c[1], c[2], _S.act_scale, 0)

    def conv(self, x, bias, pre_bn=None, residual=None, layout=0):
        """conv(relu(bn(x))) + bias + residual (``pre_bn`` = (scale, shift) of the BN in front, or None).
        On a device 3x3 layers run on the matrix cores (split-f16 implicit GEMM of csrc/conv.hip, BN +
        ReLU fused into its prologue, bias / residual into its epilogue) and so do the 1x1 skips; CPU
        tensors (validation against the reference classes, inside nets.cpu_reference()) take the torch
        composition."""
        if self.k == 3 and _fused_ok(x, *([] if residual is None else [residual])):
            cout, cin = self.weight.shape[:2]
            N, _, H, W = x.shape
            buf, wscale, xscale, arith = self._split_weights()
            out = torch.empty(N, cout, H, W, device=x.device, dtype=x.dtype)
            sc, sh = pre_bn if pre_bn is not None else (None, None)
            with torch.cuda.device(x.device):
                _lib.check(_lib.lib().slr_conv3x3_forward(_lib.ptr(x), _lib.ptr(buf), _lib.ptr(bias), _lib.ptr(residual), _lib.ptr(out),
                                                          N, cin, cout, H, W, wscale, xscale, _lib.ptr(sc), _lib.ptr(sh),
                                                          layout | arith, _lib.stream_of(x)), "slr_conv3x3_forward")
            return out
        assert layout == 0 or x.is_cuda        # the channel-blocked intermediate exists on the device path only
        if residual is not None:
            return self.conv(x, bias, pre_bn) + residual
        if pre_bn is not None:
            x = bn_relu_mask(x, pre_bn[0], pre_bn[1], False)
        if self.k == 1 and self.weight.shape[0] <= 4 and ((layout & IN_B8) or (x.shape[2] * x.shape[3]) % 4 == 0) \
                and _fused_ok(x):
            N, cin, H, W = x.shape                      # skip branch onto the 3 output channels: HBM-bound HIP kernel
            cout = self.weight.shape[0]
            out = torch.empty(N, cout, H, W, device=x.device, dtype=x.dtype)
            with torch.cuda.device(x.device):
                assert not (layout & OUT_B8)
                _lib.check(_lib.lib().slr_conv1x1_small(_lib.ptr(x), _lib.ptr(self.weight), _lib.ptr(bias), _lib.ptr(out),
                                                        N, cin, cout, H, W, int(bool(layout & IN_B8)), _lib.stream_of(x)),
                           "slr_conv1x1_small")
            return out
        if self.k == 1 and _fused_ok(x):                 # the other 1x1 skip branches: split-f16 MFMA, HBM-bound
            N, cin, H, W = x.shape
            cout = self.weight.shape[0]
            buf, wscale, xscale, arith = self._split_weights()
            out = torch.empty(N, cout, H, W, device=x.device, dtype=x.dtype)
            with torch.cuda.device(x.device):
                _lib.check(_lib.lib().slr_conv1x1_forward(_lib.ptr(x), _lib.ptr(buf), _lib.ptr(bias), _lib.ptr(out),
                                                          N, cin, cout, H, W, wscale, xscale, layout | arith, _lib.stream_of(x)),
                           "slr_conv1x1_forward")
            return out
        return F.conv2d(x, self.weight, bias, padding=self.pad)


class PartialConv(Conv):
    """PartialConv2d(multi_channel=True, return_mask=True), models/layers/partialconv2d.py:41-81,
    together with the BN + ReLU + input*mask in front of it (blocks.py:229-236).

    ``mask``: [N,1,H,W] channel-uniform mask, or None = (x != 0) per element
    (models/networks/architectures.py:369; needs ``pre_bn``).  ``pre_bn`` = (scale, shift): the input
    of the convolution is relu(x*scale - shift)*mask; None: ``x`` is already that tensor (the output
    of the previous partial convolution called with ``next_bn``).
    conv(mask, ones[out,in,k,k]) (:61) is the same k x k box sum for every output channel:
    box_k(mplane)*mscale with (mplane, mscale) = (mask, Cin) or (channel sum of the per-element
    mask, 1) -- exact integer arithmetic in fp32.
    Returns (out, update_mask [N,1,H,W]); with ``next_bn`` the output is already the activated,
    masked input of the block's second convolution.  On a device all of this is ONE kernel
    (slr_pconv3x3_forward, which also forms the box sum and the per-element mask from the staged
    input); on the CPU the torch composition that defines it."""

    def forward_skipout(self, x, mask, skip_conv, next_bn, pre_bn, layout=0):
        """The block's first partial convolution and, from the same pass over x, the block's 1x1 skip convolution of the raw x: Cout <= 4,
        channel-blocked x (slr_pconv3x3_forward_skipout; blocks.py:229-236, 243-247).  Returns (out, update_mask, skip)."""
        assert self.k == 3 and skip_conv.k == 1 and skip_conv.bias is None and _fused_ok(x, *([] if mask is None else [mask]))
        cin = x.shape[1]
        N, _, H, W = x.shape
        cout = self.weight.shape[0]
        buf, wscale, xscale, arith = self._split_weights()
        out = torch.empty(N, cout, H, W, device=x.device, dtype=x.dtype)
        skip = torch.empty(N, cout, H, W, device=x.device, dtype=x.dtype)
        um = torch.empty(N, 1, H, W, device=x.device, dtype=x.dtype)
        psc, psh = pre_bn if pre_bn is not None else (None, None)
        nsc, nsh = next_bn if next_bn is not None else (None, None)
        with torch.cuda.device(x.device):
            _lib.check(_lib.lib().slr_pconv3x3_forward_skipout(
                _lib.ptr(x), _lib.ptr(psc), _lib.ptr(psh), _lib.ptr(mask), _lib.ptr(buf), wscale, xscale, _lib.ptr(self.bias), None,
                _lib.ptr(nsc), _lib.ptr(nsh), _lib.ptr(out), _lib.ptr(um), N, cin, cout, H, W,
                _lib.ptr(skip_conv._w4()), _lib.ptr(skip), layout | arith, _lib.stream_of(x)), "slr_pconv3x3_forward_skipout")
        return out, um, skip

    def forward_skip(self, x, mask, skip_x, skip_conv, layout=0, skip_b8=False, pool=False):
        """The block's second partial convolution with the 1x1 skip branch inside (slr_pconv3x3_forward_skip; blocks.py:237-248):
        ``x`` is the activated, masked output of the first one.  Returns (out, update_mask)."""
        assert self.k == 3 and skip_conv.k == 1 and skip_conv.bias is None and _fused_ok(x, mask, skip_x)
        cin = x.shape[1]
        N, _, H, W = x.shape
        cout = self.weight.shape[0]
        buf, wscale, xscale, arith = self._split_weights()
        sbuf, swscale, _, sarith = skip_conv._split_weights()
        assert arith == sarith and not (arith & CONV_WINO)
        out, ws, ws_bytes, pflag = _pool_out(N, cout, H, W, x, pool)
        um = torch.empty(N, 1, H, W, device=x.device, dtype=x.dtype)
        with torch.cuda.device(x.device):
            _lib.check(_lib.lib().slr_pconv3x3_forward_skip(
                _lib.ptr(x), None, None, _lib.ptr(mask), _lib.ptr(buf), wscale, xscale, _lib.ptr(self.bias), _lib.ptr(out), _lib.ptr(um),
                N, cin, cout, H, W, _lib.ptr(skip_x), _lib.ptr(sbuf), skip_x.shape[1], swscale, _lib.ptr(ws), ws_bytes,
                layout | arith | pflag | (SKIP_B8 if skip_b8 else 0), _lib.stream_of(x)), "slr_pconv3x3_forward_skip")
        return out, um

    def forward(self, x, mask, residual=None, next_bn=None, pre_bn=None, layout=0):
        cin = x.shape[1]
        assert mask is not None or pre_bn is not None
        if self.k == 3 and _fused_ok(x, *([] if mask is None else [mask]), *([] if residual is None else [residual])):
            N, _, H, W = x.shape
            cout = self.weight.shape[0]
            buf, wscale, xscale, arith = self._split_weights()
            out = torch.empty(N, cout, H, W, device=x.device, dtype=x.dtype)
            um = torch.empty(N, 1, H, W, device=x.device, dtype=x.dtype)
            psc, psh = pre_bn if pre_bn is not None else (None, None)
            nsc, nsh = next_bn if next_bn is not None else (None, None)
            with torch.cuda.device(x.device):
                _lib.check(_lib.lib().slr_pconv3x3_forward(
                    _lib.ptr(x), _lib.ptr(psc), _lib.ptr(psh), _lib.ptr(mask), _lib.ptr(buf), wscale, xscale,
                    _lib.ptr(self.bias), _lib.ptr(residual), _lib.ptr(nsc), _lib.ptr(nsh), _lib.ptr(out), _lib.ptr(um),
                    N, cin, cout, H, W, layout | arith, _lib.stream_of(x)), "slr_pconv3x3_forward")
            return out, um
        assert layout == 0
        if mask is None:
            mplane, mscale = (x != 0).sum(1, keepdim=True).to(x.dtype), 1.0
        else:
            mplane, mscale = mask, float(cin)
        box = F.avg_pool2d(mplane, self.k, stride=1, padding=self.pad, divisor_override=1)
        xin = bn_relu_mask(x, pre_bn[0], pre_bn[1], mask) if pre_bn is not None else x
        raw0 = self.conv(xin, None)                                                # bias joins in the epilogue
        return pconv_epilogue(raw0, self.bias, box, mscale, cin * self.k * self.k, residual, next_bn)


def avgpool_down(x, b8=False):
    """nn.AvgPool2d(3, stride=2, padding=1), blocks.py:196-199 (b8: x and the result are channel-blocked)."""
    if _fused_ok(x):
        N, C, H, W = x.shape
        out = torch.empty(N, C, (H - 1) // 2 + 1, (W - 1) // 2 + 1, device=x.device, dtype=x.dtype)
        with torch.cuda.device(x.device):
            _lib.check(_lib.lib().slr_avgpool3x3s2(_lib.ptr(x), _lib.ptr(out), N, C, H, W, int(b8), _lib.stream_of(x)),
                       "slr_avgpool3x3s2")
        return out
    return F.avg_pool2d(x, 3, stride=2, padding=1)


def upsample_up(x, b8=False):
    """nn.Upsample(scale_factor=2, mode='bilinear'), blocks.py:200-203 (b8: x and the result are channel-blocked)."""
    if _fused_ok(x):
        N, C, H, W = x.shape
        out = torch.empty(N, C, 2 * H, 2 * W, device=x.device, dtype=x.dtype)
        with torch.cuda.device(x.device):
            _lib.check(_lib.lib().slr_upsample_bilinear2x(_lib.ptr(x), _lib.ptr(out), N, C, H, W, int(b8), _lib.stream_of(x)),
                       "slr_upsample_bilinear2x")
        return out
    return F.interpolate(x, scale_factor=2, mode="bilinear", align_corners=False)


def _resample(kind):
    if kind == "Up":
        return upsample_up
    if kind:                                                       # "Down" / encoder "downsample=True"
        return avgpool_down
    return lambda x, b8=False: x


def _resample_mask(kind):
    if kind == "Down":
        return lambda m: F.max_pool2d(m, 3, stride=2, padding=1)
    if kind == "Up":
        return lambda m: F.interpolate(m, scale_factor=2, mode="nearest")
    return lambda m: m


class ResBlock(nn.Module):
    """ResNet_Block, models/layers/blocks.py:47-87."""

    def __init__(self, cin, cout, resample=None):
        super().__init__()
        self.bn1, self.bn2 = AffineBN(cin), AffineBN(cout)
        self.conv_aa, self.conv_ab = Conv(cin, cout, 3), Conv(cout, cout, 3)
        self.conv_b = Conv(cin, cout, 1) if (resample or cin != cout) else None
        self.resample = _resample(resample)
        self.pools = "Up" if resample == "Up" else bool(resample)        # the resampling the fused second convolution can take into its epilogue

    def forward(self, x, b8_in=False):
        """-> (y, b8_out): ``b8_in`` / ``b8_out`` = x / y are channel-blocked in memory (see _b8)."""
        cout = self.conv_aa.weight.shape[0]
        b8 = _b8(x, cout)                                   # layout of everything this block produces
        lin = IN_B8 if b8_in else 0
        if self.conv_b is not None and cout <= 4 and b8_in and _S.fused_skips and x.is_cuda and not _S.torch_convs:
            # the narrow end (128 -> 3): conv_aa and the skip conv_b(x) from one pass over x
            a, b = self.conv_aa.forward_skipout(x, self.bn1.scale_shift(), self.conv_b, layout=lin)
            a = self.conv_ab(a, self.bn2.scale_shift(), residual=b)
            return self.resample(a, b8), b8
        a = self.conv_aa(x, self.bn1.scale_shift(), layout=lin | (OUT_B8 if b8 else 0))   # BN + ReLU ride in the prologue
        if self.conv_b is not None and _skip_rides(a, cout, b8, b8_in):                        # x_a + conv_b(x) (:83-87) in one kernel
            pool = self.pools if (cout > 64 and _S.fused_pools) else False
            a = self.conv_ab.forward_skip(a, self.bn2.scale_shift(), x, self.conv_b, layout=IN_B8 | OUT_B8, skip_b8=b8_in, pool=pool)
            return (a if pool else self.resample(a, b8)), b8
        if self.conv_b is not None:
            skip_b8 = b8 and cout > 4                       # (the <= 4-channel skip kernel writes NCHW)
            b = self.conv_b(x, layout=lin | (OUT_B8 if skip_b8 else 0))
        else:
            b, skip_b8 = x, b8_in
        a = self.conv_ab(a, self.bn2.scale_shift(), residual=b,                           # x_a + x_b (:87) in the epilogue
                         layout=(IN_B8 if b8 else 0) | (OUT_B8 if b8 else 0) | (RES_B8 if skip_b8 and b8 else 0))
        return self.resample(a, b8), b8       # == resample(x_a) + resample(x_b): both resamplers are linear


class PconvResBlock(nn.Module):
    """ResNet_Block_Pconv2 with pconv_pbn_woresbias, models/layers/blocks.py:173-248."""

    def __init__(self, cin, cout, resample=None):
        super().__init__()
        self.bn1, self.bn2 = AffineBN(cin), AffineBN(cout)
        self.conv_aa, self.conv_ab = PartialConv(cin, cout, 3), PartialConv(cout, cout, 3)
        self.conv_b = Conv(cin, cout, 1, bias=False) if (resample or cin != cout) else None   # :192-193
        self.resample, self.resample_mask = _resample(resample), _resample_mask(resample)
        self.has_resample = bool(resample)
        self.pools = "Up" if resample == "Up" else bool(resample)        # the resampling the fused second convolution can take into its epilogue

    def forward(self, x, mask, b8_in=False):
        """-> (y, update_mask, b8_out).  mask: None = (x != 0) per channel (architectures.py:369; x is NCHW then), else
        [N,1,H,W] channel-uniform; ``b8_in`` / ``b8_out``: x / y are channel-blocked in memory (see _b8)."""
        cout = self.conv_aa.weight.shape[0]
        b8 = _b8(x, cout)
        lin = IN_B8 if b8_in else 0
        if self.conv_b is not None and cout <= 4 and b8_in and mask is not None and _S.fused_skips and x.is_cuda and not _S.torch_convs:
            # the narrow end (128 -> 3): conv_aa and the skip conv_b(x) from one pass over x
            a, m, skip = self.conv_aa.forward_skipout(x, mask, self.conv_b, self.bn2.scale_shift(), self.bn1.scale_shift(), layout=lin)
            a, m = self.conv_ab(a, m, residual=skip)
            return self.resample(a, b8), self.resample_mask(m), b8
        a, m = self.conv_aa(x, mask, next_bn=self.bn2.scale_shift(), pre_bn=self.bn1.scale_shift(),
                            layout=lin | (OUT_B8 if b8 else 0))                      # :229-236
        # x_a + x_b (:248).  The reference resamples the two branches separately and adds; avg-pool
        # and bilinear up-sampling are linear, so resample(x_a + x_b) is the same result up to fp32
        # rounding, lets the residual join the epilogue, and halves the resampling work.
        if self.conv_b is not None and _skip_rides(a, cout, b8, b8_in):                # :237-248 in one kernel
            pool = self.pools if (cout > 64 and _S.fused_pools) else False
            a, m = self.conv_ab.forward_skip(a, m, x, self.conv_b, layout=IN_B8 | OUT_B8, skip_b8=b8_in, pool=pool)
            return (a if pool else self.resample(a, b8)), self.resample_mask(m), b8
        if self.conv_b is not None:                                                # :243-247
            skip_b8 = b8 and cout > 4
            skip = self.conv_b(x, layout=lin | (OUT_B8 if skip_b8 else 0))
        else:
            skip, skip_b8 = x, b8_in
        a, m = self.conv_ab(a, m, residual=skip,                                   # :237-239
                            layout=(IN_B8 if b8 else 0) | (OUT_B8 if b8 else 0) | (RES_B8 if skip_b8 and b8 else 0))
        return self.resample(a, b8), self.resample_mask(m), b8                     # :240-241


# --------------------------------------------------------------------------- networks

_ENC = [3, 32, 32, 32, 64, 64, 64, 64]                      # configs.py:96-106 (ngf = 64)
_DEC = [64, 128, 256, 256, 128, 128, 128]                   # configs.py:117-127, inner widths
_UPDOWN = [False, "Down", "Down", False, "Up", "Up", False, False]   # configs.py:128-137


class EncoderWithZ(nn.Module):
    """ResNetEncoder_with_Z, models/networks/architectures.py:155-197: 8 blocks, no down-sampling,
    last block emits 64 features + 1 Z channel."""

    def __init__(self, cin=3, feat=64):
        super().__init__()
        ch = [cin] + _ENC[1:] + [feat + 1]
        self.blocks = nn.ModuleList(ResBlock(ch[i], ch[i + 1]) for i in range(8))

    def forward(self, x):
        b8 = False
        for b in self.blocks:
            x, b8 = b(x, b8)
        assert not b8                                    # 65 channels: NCHW
        return x[:, :-1].contiguous(), x[:, -1:].contiguous()          # :195-197


class Encoder(nn.Module):
    """ResNetEncoder (alpha encoder of SLR v1: 3 -> ... -> 2), architectures.py:121-153."""

    def __init__(self, cin=3, cout=2):
        super().__init__()
        ch = [cin] + _ENC[1:] + [cout]
        self.blocks = nn.ModuleList(ResBlock(ch[i], ch[i + 1]) for i in range(8))

    def forward(self, x):
        b8 = False
        for b in self.blocks:
            x, b8 = b(x, b8)
        assert not b8
        return x


class DecoderPconv2(nn.Module):
    """ResNetDecoderPconv2, architectures.py:345-375: input mask = (x != 0) (:369)."""

    def __init__(self, cin=64, cout=3):
        super().__init__()
        ch = [cin] + _DEC + [cout]
        self.blocks = nn.ModuleList(PconvResBlock(ch[i], ch[i + 1], _UPDOWN[i]) for i in range(8))

    def forward(self, x):
        mask, b8 = None, False                           # (x != 0) per channel, derived inside the first block
        for b in self.blocks:
            x, mask, b8 = b(x, mask, b8)
        assert not b8                                    # the 1- / 3-channel end is NCHW
        return x


class BGDecoder(nn.Module):
    """ResNetBGDecoder (net_bg of SLR v1), architectures.py:233-260, arch 256W8UpDown64BG."""

    def __init__(self, cin=3, cout=3):
        super().__init__()
        ch = [cin] + _DEC + [cout]
        self.blocks = nn.ModuleList(ResBlock(ch[i], ch[i + 1], _UPDOWN[i]) for i in range(8))

    def forward(self, x):
        b8 = False
        for b in self.blocks:
            x, b8 = b(x, b8)
        assert not b8
        return x


def saturation_count(device, reset=True):
    """Waves of the split-f16 kernels on ``device`` that had to clamp an activation since the last reset (|x| beyond the
    exact domain of the split, csrc/conv.hip).  The counter is one per device; the read is ordered on torch's current
    stream of that device and synchronises it with the host."""
    import ctypes
    n = ctypes.c_ulonglong(0)
    with torch.cuda.device(device):
        _lib.check(_lib.lib().slr_conv_saturation_count(ctypes.byref(n), 1 if reset else 0,
                                                        ctypes.c_void_p(torch.cuda.current_stream(device).cuda_stream)),
                   "slr_conv_saturation_count")
    return int(n.value)


def check_saturation(device, what="convolution", reset=True):
    """Raise if a split-f16 kernel on ``device`` had to clamp an activation since the last check.  The result of such a
    launch is wrong, not merely inexact, and the reference's fp32 convolution has no such limit -- so it is an error,
    never silent.  (The animators do better than raising: pipeline.py, convs="auto".)"""
    n = saturation_count(device, reset)
    if n:
        raise RuntimeError(f"slr_sfs_amd: {n} wave(s) of the {what} kernels met activations >= {65472.0 / _S.act_scale:.0f} in "
                           f"magnitude, outside the exact range of the split-f16 matrix-core convolution; the result is not "
                           f"valid (use a smaller nets.activation_scale, nets.fp32_kernels(), or the animators' "
                           f"convs='auto')")
    return 0


class SaturationLog:
    """Asynchronous per-piece saturation records of one clip: ``mark()`` after a piece of work (a decoder batch) copies
    the device counter into pinned host memory in stream order -- no host synchronisation; ``bad()`` (after the stream
    has been synchronised) lists the pieces during which the counter moved."""

    def __init__(self, device, capacity):
        self.device = device
        self.slots = torch.zeros(capacity + 1, dtype=torch.int32).pin_memory()
        self.n = 0
        self._record()                                   # the value the clip starts from

    def _record(self):
        import ctypes
        with torch.cuda.device(self.device):
            _lib.check(_lib.lib().slr_conv_saturation_record(ctypes.c_void_p(self.slots.data_ptr() + 4 * self.n),
                                                             ctypes.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)),
                       "slr_conv_saturation_record")
        self.n += 1

    def mark(self):
        self._record()

    def bad(self):
        torch.cuda.current_stream(self.device).synchronize()
        v = self.slots[:self.n].tolist()
        return [i for i in range(self.n - 1) if v[i + 1] != v[i]]


def guarded(fn, device, policy, what, owner=None):
    """Run ``fn()`` (networks on split-f16 kernels) under the saturation policy of the animators:
    "split": as is, raise if an activation was clamped; "fp32" / "fp32-winograd": inside fp32_kernels() (direct / Winograd 3x3);
    "auto": split-f16 at the default activation scale; clamped -> again at scale 1 (exact up to 65472); clamped again ->
    inside fp32_kernels() (the fp32 matrix instructions: no limit).  ``owner`` (an animator) remembers the rung that worked, so later clips start there.
    Synchronises with the device once per call (per rung tried)."""
    if policy in ("fp32", "fp32-winograd"):
        with fp32_kernels(winograd=policy == "fp32-winograd"):
            return fn()
    if policy == "split":
        out = fn()
        check_saturation(device, what)
        return out
    assert policy == "auto", policy
    rung = getattr(owner, "_conv_rung", 0) if owner is not None else 0
    saturation_count(device)                                   # start from a clean counter
    for r, scale in enumerate((64.0, 1.0)):
        if r < rung:
            continue
        with activation_scale(scale):
            out = fn()
        if saturation_count(device) == 0:
            return out
        import warnings
        warnings.warn(f"slr_sfs_amd: activations of the {what} exceed the exact range of the split-f16 convolutions at "
                      f"activation scale {scale:g}; rendering again " + ("at scale 1" if r == 0 else "on the fp32 rung"))
        if owner is not None:
            owner._conv_rung = r + 1
    with fp32_kernels(winograd=False):                         # (the safety net is the strict rung)
        return fn()


# --------------------------------------------------------------------------- checkpoints

def _fold_sn(sd, key):
    """Effective eval-mode weight of a legacy spectral_norm layer: W / (u^T W_mat v)."""
    w = sd[key + ".weight_orig"] if key + ".weight_orig" in sd else sd[key + ".weight"]
    if key + ".weight_u" in sd:
        u, v = sd[key + ".weight_u"], sd[key + ".weight_v"]
        sigma = torch.dot(u, torch.mv(w.reshape(w.shape[0], -1), v))
        w = w / sigma
    return w


def _load_bn(bn, sd, key):
    """key = '...bn' or '...pbn' of a (Partial)LinearNoiseLayer; zero noise -> gain 1, bias 0."""
    bn.stored_mean.copy_(sd[key + ".stored_mean"])
    bn.stored_var.copy_(sd[key + ".stored_var"])
    bn.__dict__.pop("_ss", None)


def _load_conv(conv, sd, key):
    conv.__dict__.pop("_wsplit", None)
    conv.weight.data.copy_(_fold_sn(sd, key))
    if conv.bias is not None:
        conv.bias.data.copy_(sd[key + ".bias"])


@torch.no_grad()
def load_reference_state_dict(net, sd, prefix):
    """Fill ``net`` from a reference state dict (SURVEY App. C key scheme).  ``prefix`` e.g.
    'model.module.encoder.' / 'model.module.projector.' / 'model.module.net_bg.' ..."""
    sd = {k[len(prefix):]: v for k, v in sd.items() if k.startswith(prefix)}
    for i, blk in enumerate(net.blocks):
        if isinstance(blk, PconvResBlock):
            b = f"eblocks.{i}."
            _load_bn(blk.bn1, sd, b + "bn_noise1.pbn")
            _load_bn(blk.bn2, sd, b + "bn_noise2.pbn")
            _load_conv(blk.conv_aa, sd, b + "conv_aa")
            _load_conv(blk.conv_ab, sd, b + "conv_ab")
            if blk.conv_b is not None:
                _load_conv(blk.conv_b, sd, b + "conv_b")
        else:
            b = ("eblocks" if isinstance(net, BGDecoder) else "gblocks") + f".{i}."
            _load_bn(blk.bn1, sd, b + "ch_a.0.bn")
            _load_bn(blk.bn2, sd, b + "ch_a.3.bn")
            _load_conv(blk.conv_aa, sd, b + "ch_a.2")
            _load_conv(blk.conv_ab, sd, b + "ch_a.5")
            if blk.conv_b is not None:
                _load_conv(blk.conv_b, sd, b + "ch_b.0")
    return net
