"""Drop-in for the reference's ``models/softsplat.py`` on MI355X.

Same public names, argument meaning, assertions and error behaviour as the reference
(file:line cited per symbol, relative to the reference root); the five cupy/NVRTC CUDA kernels
(models/softsplat.py:12-326) are replaced by the HIP kernels behind include/slr_splat.h.
"""
import torch

from . import _lib
from ._lib import WS_CLEAN, check, lib, ptr, require_device, stream_of, workspace

_MODES = {"summation": 0, "average": 1, "linear": 2, "softmax": 3}


def _check_pair(input, flow):
    """Shape/contiguity contract of models/softsplat.py:393-402."""
    intFlowDepth, intFlowHeight, intFlowWidth = flow.shape[1], flow.shape[2], flow.shape[3]
    assert(intFlowDepth == 2)
    assert(input.shape[2] == intFlowHeight)
    assert(input.shape[3] == intFlowWidth)
    assert(input.shape[0] == flow.shape[0])
    assert(input.is_contiguous() == True)
    assert(flow.is_contiguous() == True)
    require_device(input, flow)


def _splat_sum(input, flow):
    N, C, H, W = input.shape
    out = torch.empty_like(input)          # every element is written by the kernel (no memset)
    ws = workspace(input, "a", N, C, H, W)
    with torch.cuda.device(input.device):
        check(lib().slr_softsplat_forward(ptr(input), ptr(flow), ptr(out), N, C, H, W,
                                          ptr(ws), ws.numel(), WS_CLEAN, stream_of(input)), "slr_softsplat_forward")
    return out


class _FunctionSoftsplat(torch.autograd.Function):
    """models/softsplat.py:388-479 -- summation splat with autograd."""

    @staticmethod
    def forward(self, input, flow):
        self.save_for_backward(input, flow)
        _check_pair(input, flow)
        return _splat_sum(input, flow)

    @staticmethod
    def backward(self, gradOutput):
        input, flow = self.saved_tensors
        assert(gradOutput.is_contiguous() == True)                      # :438
        require_device(gradOutput)
        N, C, H, W = input.shape
        gradInput = torch.empty_like(input) if self.needs_input_grad[0] == True else None
        gradFlow = torch.empty_like(flow) if self.needs_input_grad[1] == True else None
        if gradInput is not None or gradFlow is not None:
            with torch.cuda.device(input.device):
                # (channel groups -- 2-4 on the training crops, 2 on grids larger than the chip: partial gradFlow sums in a scratch tensor of torch's allocator)
                nb = int(lib().slr_softsplat_backward_ws_bytes(N, C, H, W)) if gradFlow is not None else 0
                ws = torch.empty(nb, dtype=torch.uint8, device=input.device) if nb else None
                check(lib().slr_softsplat_backward_ws(ptr(input), ptr(flow), ptr(gradOutput), ptr(gradInput),
                                                      ptr(gradFlow), N, C, H, W, ptr(ws), nb, stream_of(input)),
                      "slr_softsplat_backward_ws")
        return gradInput, gradFlow


def _needs_grad(*tensors):
    return torch.is_grad_enabled() and any(t is not None and t.requires_grad for t in tensors)


def _splat_by_composition(values, flow, metric, mode):
    """The differentiable route: one summation splat of [values * w, w] through _FunctionSoftsplat (whose backward is
    the reference's), then the division -- the arithmetic of models/softsplat.py:669-688, so gradients flow as upstream:
    w = 1 (average) | metric (linear) | exp(metric) (softmax); an exact-zero normaliser divides by 1 (:684)."""
    if mode == 'summation':
        return _FunctionSoftsplat.apply(values, flow)
    if mode == 'average':
        weight = values.new_ones(values.shape[0], 1, values.shape[2], values.shape[3])
        stacked = torch.cat([values, weight], 1)
    else:
        weight = metric.exp() if mode == 'softmax' else metric
        stacked = torch.cat([values * weight, weight], 1)
    splatted = _FunctionSoftsplat.apply(stacked, flow)
    norm = splatted[:, -1:, :, :]
    norm[norm == 0.0] = 1.0                 # in place on the view, like upstream (the zeros carry no gradient)
    return splatted[:, :-1, :, :] / norm


def FunctionSoftsplat(tenInput, tenFlow, tenMetric, strType):
    """models/softsplat.py:665-690.  Without autograd (the inference scripts run under
    torch.no_grad(), test_animating/test_baseline_4eval_rawsize.py:244) weighting, splat and
    normalisation run as ONE fused kernel; with autograd the reference's composition around
    _FunctionSoftsplat is kept so gradients flow exactly as upstream."""
    assert(tenMetric is None or tenMetric.shape[1] == 1)
    assert(strType in ['summation', 'average', 'linear', 'softmax'])

    if strType == 'summation' or _needs_grad(tenInput, tenFlow, tenMetric):
        return _splat_by_composition(tenInput, tenFlow, tenMetric, strType)

    # The reference only asserts contiguity of what reaches _FunctionSoftsplat, and in these modes that is the result
    # of torch.cat (:669-676), always contiguous: channel slices / permuted tensors are accepted like upstream.
    tenInput = tenInput.contiguous()
    if tenMetric is not None:
        tenMetric = tenMetric.contiguous()
    _check_pair(tenInput, tenFlow)
    if strType != 'average':
        require_device(tenMetric)
        assert(tenMetric.shape[0] == tenInput.shape[0] and tenMetric.shape[2:] == tenInput.shape[2:])
    N, C, H, W = tenInput.shape
    out = torch.empty_like(tenInput)
    ws = workspace(tenInput, "a", N, C, H, W)
    with torch.cuda.device(tenInput.device):
        check(lib().slr_softsplat_mode_forward(ptr(tenInput), ptr(tenMetric) if strType != 'average' else None,
                                               ptr(tenFlow), ptr(out), N, C, H, W, _MODES[strType],
                                               ptr(ws), ws.numel(), WS_CLEAN, stream_of(tenInput)),
              "slr_softsplat_mode_forward")
    return out


class ModuleSoftsplat(torch.nn.Module):
    """models/softsplat.py:692-702."""

    def __init__(self, strType):
        super(ModuleSoftsplat, self).__init__()
        self.strType = strType

    def forward(self, tenInput, tenFlow, tenMetric):
        return FunctionSoftsplat(tenInput, tenFlow, tenMetric, self.strType)


def _maxsplat(input, flow, init):
    _check_pair(input, flow)
    N, C, H, W = input.shape
    out = torch.empty_like(input)
    ws = workspace(input, "a", N, C, H, W)
    with torch.cuda.device(input.device):
        check(lib().slr_maxsplat_forward(ptr(input), ptr(flow), ptr(out), float(init), N, C, H, W,
                                         ptr(ws), ws.numel(), WS_CLEAN, stream_of(input)), "slr_maxsplat_forward")
    return out


class _FunctionMaximumsplat(torch.autograd.Function):
    """models/softsplat.py:482-518 -- forward only, like the reference (no backward defined)."""

    @staticmethod
    def forward(self, input, flow):
        self.save_for_backward(input, flow)
        return _maxsplat(input, flow, 0.0)                               # new_zeros, :497


def _FunctionMaximumWarpNormsplat(input, flow):
    """models/softsplat.py:576-624 -- max-splat seeded with -1000 (:590), gathered back per source
    pixel together with the input itself (:606-618)."""
    _check_pair(input, flow)
    N, C, H, W = input.shape
    scratch = torch.empty_like(input)
    out = torch.empty_like(input)
    ws = workspace(input, "a", N, C, H, W)
    with torch.cuda.device(input.device):
        check(lib().slr_max_warp_norm(ptr(input), ptr(flow), ptr(scratch), ptr(out), N, C, H, W,
                                      ptr(ws), ws.numel(), WS_CLEAN, stream_of(input)), "slr_max_warp_norm")
    return out


class ModuleMaximumsplat(torch.nn.Module):
    """models/softsplat.py:705-712."""

    def __init__(self):
        super(ModuleMaximumsplat, self).__init__()

    def forward(self, tenInput, tenFlow):
        return _FunctionMaximumsplat.apply(tenInput, tenFlow)


class ModuleMaximumWarpNormsplat(torch.nn.Module):
    """models/softsplat.py:716-724."""

    def __init__(self):
        super(ModuleMaximumWarpNormsplat, self).__init__()

    def forward(self, tenInput, tenFlow):
        return _FunctionMaximumWarpNormsplat(tenInput, tenFlow)


def splat_normalize(accum, norm_mode="zero_to_one", eps=1e-8):
    """accum [N,C+1,H,W] (last channel = normaliser) -> [N,C,H,W].
    'zero_to_one': models/softsplat.py:681-686; 'clamp': animating_softmax_splating.py:923-924."""
    require_device(accum)
    N, C1, H, W = accum.shape
    out = accum.new_empty(N, C1 - 1, H, W)
    with torch.cuda.device(accum.device):
        check(lib().slr_splat_normalize(ptr(accum), ptr(out), N, C1 - 1, H, W,
                                        0 if norm_mode == "zero_to_one" else 1, float(eps), stream_of(accum)),
              "slr_splat_normalize")
    return out
