"""CPU oracle for the SLR-SFS frame-synthesis hot path -- TEST INFRASTRUCTURE ONLY.

ctypes front-end of ``oracle/liboracle.so`` (built from ``oracle/slr_oracle.c`` by
``oracle/Makefile``) plus numpy restatements of the reference's host-side glue around the
kernels.  Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg
may import this module; the product package ``slr_sfs_amd`` never does.

Parity status: PINNED against ``tests/golden/*.npz`` (see slr_oracle.c header and
tools/make_golden.py).  All citations are file:line under /root/reference.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None
_F = ctypes.POINTER(ctypes.c_float)


def build(force=False):
    """Compile liboracle.so with gcc (seconds)."""
    so = os.path.join(_HERE, "liboracle.so")
    src = os.path.join(_HERE, "slr_oracle.c")
    if force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-B", "liboracle.so"],
                              stdout=subprocess.DEVNULL)
    return so


def lib():
    global _LIB
    if _LIB is None:
        so = os.path.join(_HERE, "liboracle.so")
        if not os.path.exists(so):
            build()
        _LIB = ctypes.CDLL(so)
        i = ctypes.c_int
        _LIB.oracle_euler_integrate.argtypes = [_F, i, i, i, _F, _F]
        _LIB.oracle_euler_integrate_all.argtypes = [_F, i, i, i, _F, _F]
        _LIB.oracle_euler_backward.argtypes = [_F, i, i, i, _F, _F]
        _LIB.oracle_euler_backward.restype = None
        _LIB.oracle_softsplat_forward.argtypes = [_F, _F, _F, i, i, i, i]
        _LIB.oracle_softsplat_grad_input.argtypes = [_F, _F, _F, i, i, i, i]
        _LIB.oracle_softsplat_grad_flow.argtypes = [_F, _F, _F, _F, i, i, i, i]
        _LIB.oracle_maxsplat_forward.argtypes = [_F, _F, _F, i, i, i, i]
        _LIB.oracle_inversesplat.argtypes = [_F, _F, _F, i, i, i, i]
        _LIB.oracle_set_threads.argtypes = [i]
        _LIB.oracle_max_threads.restype = i
        for f in ("oracle_euler_integrate", "oracle_euler_integrate_all", "oracle_softsplat_forward",
                  "oracle_softsplat_grad_input", "oracle_softsplat_grad_flow",
                  "oracle_maxsplat_forward", "oracle_inversesplat", "oracle_set_threads"):
            getattr(_LIB, f).restype = None
    return _LIB


def set_threads(n):
    lib().oracle_set_threads(int(n))


def max_threads():
    return int(lib().oracle_max_threads())


def _c(a):
    a = np.ascontiguousarray(a, dtype=np.float32)
    return a


def _p(a):
    return a.ctypes.data_as(_F)


# ----------------------------------------------------------------------------- euler

def euler_integration(motion, n):
    """euler_integration(motion, n) -- euler_integration_manipulator.py:7-56.
    motion [1,2,H,W] -> (disp [1,2,H,W], visible [1,1,H,W])."""
    motion = _c(motion)
    assert motion.ndim == 4 and motion.shape[0] == 1 and motion.shape[1] == 2
    H, W = motion.shape[2:]
    disp = np.empty((1, 2, H, W), np.float32)
    vis = np.empty((1, 1, H, W), np.float32)
    lib().oracle_euler_integrate(_p(motion), H, W, int(n), _p(disp), _p(vis))
    return disp, vis


def euler_integration_all(motion, nmax):
    """All frames t=0..nmax: out[t] == euler_integration(motion, t).
    -> (disp_all [nmax+1,2,H,W], vis_all [nmax+1,1,H,W])."""
    motion = _c(motion)
    H, W = motion.shape[2:]
    disp = np.empty((nmax + 1, 2, H, W), np.float32)
    vis = np.empty((nmax + 1, 1, H, W), np.float32)
    lib().oracle_euler_integrate_all(_p(motion), H, W, int(nmax), _p(disp), _p(vis))
    return disp, vis


def euler_backward(motion, n, grad_disp):
    """Gradient of euler_integration(motion, n)[0] w.r.t. motion (torch autograd through
    euler_integration_manipulator.py:36-55).  motion, grad_disp [1,2,H,W] -> grad_motion [1,2,H,W]."""
    motion, grad_disp = _c(motion), _c(grad_disp)
    H, W = motion.shape[2:]
    gm = np.empty((1, 2, H, W), np.float32)
    lib().oracle_euler_backward(_p(motion), H, W, int(n), _p(grad_disp), _p(gm))
    return gm


# ----------------------------------------------------------------------------- splat

def softsplat_forward(inp, flow):
    """_FunctionSoftsplat.forward -- softsplat.py:157-202,390-424 (summation splat)."""
    inp, flow = _c(inp), _c(flow)
    N, C, H, W = inp.shape
    assert flow.shape == (N, 2, H, W)
    out = np.empty_like(inp)
    lib().oracle_softsplat_forward(_p(inp), _p(flow), _p(out), N, C, H, W)
    return out


def softsplat_backward(inp, flow, gout):
    """_FunctionSoftsplat.backward -- softsplat.py:204-326,427-478 -> (gradInput, gradFlow)."""
    inp, flow, gout = _c(inp), _c(flow), _c(gout)
    N, C, H, W = inp.shape
    gin = np.empty_like(inp)
    gflow = np.empty_like(flow)
    lib().oracle_softsplat_grad_input(_p(flow), _p(gout), _p(gin), N, C, H, W)
    lib().oracle_softsplat_grad_flow(_p(inp), _p(flow), _p(gout), _p(gflow), N, C, H, W)
    return gin, gflow


def maxsplat_forward(inp, flow, init=0.0):
    """_FunctionMaximumsplat.forward -- softsplat.py:12-82,482-518 (init 0, :497)."""
    inp, flow = _c(inp), _c(flow)
    N, C, H, W = inp.shape
    out = np.full_like(inp, np.float32(init))
    lib().oracle_maxsplat_forward(_p(inp), _p(flow), _p(out), N, C, H, W)
    return out


def maximum_warp_norm_splat(inp, flow):
    """_FunctionMaximumWarpNormsplat -- softsplat.py:576-624: max-splat seeded with -1000
    (:590), then inverse-splat seeded with input.clone() (:606)."""
    inp, flow = _c(inp), _c(flow)
    N, C, H, W = inp.shape
    mw = maxsplat_forward(inp, flow, init=-1000.0)
    out = inp.copy()
    lib().oracle_inversesplat(_p(mw), _p(flow), _p(out), N, C, H, W)
    return out


def function_softsplat(inp, flow, metric, mode):
    """FunctionSoftsplat(tenInput, tenFlow, tenMetric, strType) -- softsplat.py:665-690."""
    assert metric is None or metric.shape[1] == 1                       # :666
    assert mode in ("summation", "average", "linear", "softmax")        # :667
    inp = _c(inp)
    if mode == "average":                                               # :669-670
        inp = np.concatenate([inp, np.ones_like(inp[:, :1])], 1)
    elif mode == "linear":                                              # :672-673
        inp = np.concatenate([inp * metric, metric], 1)
    elif mode == "softmax":                                             # :675-676
        e = np.exp(_c(metric))
        inp = np.concatenate([inp * e, e], 1)
    out = softsplat_forward(inp, flow)                                  # :680
    if mode != "summation":                                             # :681-686
        norm = out[:, -1:].copy()
        norm[norm == 0.0] = 1.0
        out = out[:, :-1] / norm
    return out


# ----------------------------------------------------------- forward_flow data-flow (a6)

def _exp32(x):
    return np.exp(x.astype(np.float32)).astype(np.float32)


def synth_baseline(fs, Z, motion, t, N, clamp_z=None, variant=None):
    """Decoder input of AnimatingSoftmaxSplating.forward_flow for index=[0,t,N-1]
    -- animating_softmax_splating.py:847-862,884-924.  fs [1,64,H,W], Z [1,1,H,W],
    motion [1,2,H,W] -> gen_fs [1,64,H,W].  clamp_z=None is the shipped behaviour
    (SURVEY App. A-5); (lo,hi) reproduces :856-859."""
    fs, Z, motion = _c(fs), _c(Z), _c(motion)
    f32 = np.float32
    disp_f, _ = euler_integration(motion, t)                            # :847  m - s
    disp_p, _ = euler_integration(-motion, N - t)                       # :848  e - m + 1
    if variant == "v2":                                                 # :849-851
        Zn = Z - maximum_warp_norm_splat(Z, disp_f)
    elif variant == "v1":                                               # :852-853
        Zn = Z
    else:
        Zn = Z - Z.max()                                                # :855
    if clamp_z is not None:
        Zn = np.clip(Zn, f32(clamp_z[0]), f32(clamp_z[1]))              # :859
    alpha = f32(1.0) - f32(t) / f32(N)                                  # :860  1 - (m-s)/(e-s+1)
    e = _exp32(Zn)
    in_f = np.concatenate([fs * e * alpha, e * alpha], 1)               # :862
    in_p = np.concatenate([fs * e * (f32(1.0) - alpha), e * (f32(1.0) - alpha)], 1)   # :895
    S = softsplat_forward(in_f, disp_f)                                 # :884-887
    gen = S[:, :-1].copy()
    norm = S[:, -1:].copy()
    Sp = softsplat_forward(in_p, disp_p)                                # :916-919
    gen += Sp[:, :-1]                                                   # :920
    norm += Sp[:, -1:]                                                  # :921
    norm = np.maximum(norm, f32(1e-8))                                  # :923
    return gen / norm                                                   # :924


def _sigmoid32(x):
    return (1.0 / (1.0 + np.exp(-x.astype(np.float32)))).astype(np.float32)


def synth_v1(fs, Z, alpha_fluid_logit, alpha_bg, motion, t, N, use_alpha0=True, variant=None):
    """Decoder inputs of AnimatingSoftmaxSplatingJoint.forward_flow for index=[0,t,N-1]
    -- animating_softmax_splating_2layers_alpha_seperate.py:921-922,950-1045.
    alpha_fluid_logit = alpha_output[:,1:2] (:944), alpha_bg = sigmoid(alpha_output[:,0:1]) (:946).
    -> (gen_fs [1,64,H,W], alpha_fluid [1,1,H,W], alpha_fluid_mask [1,1,H,W])."""
    fs, Z, motion = _c(fs), _c(Z), _c(motion)
    af, abg = _c(alpha_fluid_logit), _c(alpha_bg)
    f32 = np.float32
    disp_f, _ = euler_integration(motion, t)                            # :921
    disp_p, _ = euler_integration(-motion, N - t)                       # :922
    alpha = f32(1.0) - f32(t) / f32(N)                                  # :950
    alpha = np.clip(alpha, f32(1.0 / 600.0), f32(599.0 / 600.0))        # :952
    if variant == "v2":                                                 # :955-957
        Zn = Z - maximum_warp_norm_splat(Z, disp_f)
    elif variant == "v1":                                               # :958-959
        Zn = Z
    else:
        Zn = Z - Z.max()                                                # :961
    e = _exp32(Zn)
    if use_alpha0:                                                      # :963-972
        sg = _sigmoid32(af)
        a0 = sg / np.maximum(sg + abg, f32(1e-8))
        ea = _exp32(a0)
        pack = lambda w: np.concatenate([fs * e * w, af * ea * w, ea * w, e * w], 1)
    else:                                                               # :974-976
        pack = lambda w: np.concatenate([fs * e * w, af * e * w, e * w], 1)
    S = softsplat_forward(pack(alpha), disp_f)                          # :987-990
    S = S + softsplat_forward(pack(f32(1.0) - alpha), disp_p)           # :1024-1036
    if use_alpha0:                                                      # :992-996
        gen, afl, anorm, norm = S[:, :-3], S[:, -3:-2], S[:, -2:-1], S[:, -1:]
    else:                                                               # :998-1000
        gen, afl, norm = S[:, :-2], S[:, -2:-1], S[:, -1:]
        anorm = norm
    norm = np.maximum(norm, f32(1e-8))                                  # :1038
    mask = (norm > f32(1e-8)).astype(np.float32)                        # :1039
    gen = gen / norm                                                    # :1040
    afl = afl / (np.maximum(anorm, f32(1e-8)) if use_alpha0 else norm)  # :1041-1045
    return gen, afl, mask
