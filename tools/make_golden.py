#!/usr/bin/env python
"""Generate tests/golden/*.npz by RUNNING THE REFERENCE in this container.

Needs /root/reference (read-only) -- it does not exist on the GPU box, so only the numeric
fixtures this script writes are committed; nothing of the reference (source, bytecode,
expanded kernel text, compiled objects) is written into the repo: all intermediates live in
a temporary directory that is deleted on exit.

How the reference is executed here (no CUDA, no cupy in the image):
  * euler_integration (models/projection/euler_integration_manipulator.py:7-56) is imported
    and run UNMODIFIED on CPU; its module-level ``torch`` is replaced by a proxy that rewrites
    the hard-coded ``device='cuda'`` of three factory calls (:24-35) to 'cpu'.
  * models/softsplat.py is imported with a stub ``cupy`` module (it only needs ``memoize`` at
    import time, :383).  ``softsplat.cupy_launch`` (:383-386, the NVRTC compile+launch) is
    replaced by a host launcher: the reference's own ``cupy_kernel`` (:328-381) expands the
    kernel text for the concrete tensors, the text is compiled with g++ behind a 10-line
    prologue defining the CUDA builtins it uses (blockIdx/blockDim/threadIdx/gridDim = one
    thread, atomicAdd, atomicCAS, __float_as_int, __int_as_float), and called through ctypes.
    The grid-stride loop then visits every element sequentially -> a deterministic host
    execution of the reference kernel text.  The reference's Python (autograd Functions,
    FunctionSoftsplat, _FunctionMaximumWarpNormsplat) runs unmodified on tensors of a
    torch.Tensor subclass that reports ``is_cuda == True`` (the reference raises
    NotImplementedError for CPU tensors, :418-419).

Usage:  python tools/make_golden.py            (rewrites tests/golden/)
"""
import ctypes
import hashlib
import os
import shutil
import subprocess
import sys
import tempfile
import types

import numpy as np
import torch

REF = "/root/reference"
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")
TMP = tempfile.mkdtemp(prefix="slr_golden_")
FP64 = False        # tools/make_golden_large.py --native --fp64: the reference's kernel text compiled with every `float` a double

_PROLOGUE = r"""
#include <cmath>
#include <cstring>
struct dim3_ { int x, y, z; };
static dim3_ blockIdx = {0,0,0}, blockDim = {1,1,1}, threadIdx = {0,0,0}, gridDim = {1,1,1};
#define __global__
static inline float atomicAdd(float* p, float v) { float o = *p; *p = o + v; return o; }
static inline double atomicAdd(double* p, double v) { double o = *p; *p = o + v; return o; }
static inline int   atomicCAS(int* p, int cmp, int val) { int o = *p; if (o == cmp) *p = val; return o; }
static inline int   __float_as_int(float f) { int i; memcpy(&i, &f, 4); return i; }
static inline float __int_as_float(int i)   { float f; memcpy(&f, &i, 4); return f; }
"""


class CudaLike(torch.Tensor):
    """CPU tensor that answers is_cuda == True so the reference's GPU branch is taken."""
    is_cuda = property(lambda self: True)


def cudalike(a):
    t = torch.from_numpy(np.ascontiguousarray(a, dtype=np.float64 if FP64 else np.float32)).clone()
    return t.as_subclass(CudaLike)


def _host_launch(strFunction, strKernel):
    """Stand-in for softsplat.cupy_launch: compile the expanded kernel text for the host."""
    key = hashlib.sha1((strFunction + strKernel + ("fp64" if FP64 else "")).encode()).hexdigest()[:16]
    so = os.path.join(TMP, f"{strFunction}_{key}.so")
    if not os.path.exists(so):
        cpp = so[:-3] + ".cpp"
        if FP64:        # every `float` of the kernel text a double (the summation kernels use no bit casts)
            import re
            assert "__float_as_int" not in strKernel and "__int_as_float" not in strKernel, strFunction
            strKernel = re.sub(r"\bfloat\b", "double", strKernel)
        with open(cpp, "w") as f:
            f.write(_PROLOGUE + strKernel)
        subprocess.check_call(["g++", "-O1", "-ffp-contract=off", "-shared", "-fPIC", "-w", cpp, "-o", so])
    fn = getattr(ctypes.CDLL(so), strFunction)
    fn.restype = None

    def launch(grid, block, args):
        cargs = [ctypes.c_int(int(args[0]))] + [ctypes.c_void_p(a) for a in args[1:]]   # None -> NULL
        fn(*cargs)
    return launch


def load_reference():
    sys.path.insert(0, REF)
    cupy = types.ModuleType("cupy")
    cupy.memoize = lambda for_each_device=False: (lambda f: f)
    cupy.cuda = types.SimpleNamespace(compile_with_cache=None)
    sys.modules["cupy"] = cupy
    from models import softsplat as ss
    ss.cupy_launch = _host_launch
    import models.projection.euler_integration_manipulator as eim

    class TorchCPU:
        def __getattr__(self, n):
            a = getattr(torch, n)
            if n in ("linspace", "zeros", "ones"):
                return lambda *x, **k: a(*x, **({**k, "device": "cpu"} if k.get("device") == "cuda" else k))
            return a
    eim.torch = TorchCPU()
    return ss, eim


# ------------------------------------------------------------------ input generators

def motion_fields(H, W, rng):
    y, x = np.meshgrid(np.arange(H, dtype=np.float32), np.arange(W, dtype=np.float32), indexing="ij")
    out = {}
    u = 1.5 * np.sin(2 * np.pi * (2 * x / W + y / H) + 0.3)
    v = 1.5 * np.cos(2 * np.pi * (x / W - 1.5 * y / H) + 1.1)
    m = (x >= 0.35 * W).astype(np.float32)
    out["smooth"] = np.stack([u * m, v * m])[None].astype(np.float32)
    out["random3"] = rng.uniform(-3, 3, (1, 2, H, W)).astype(np.float32)
    # every value a multiple of 0.5 -> coordinates sit on round-half-even cliffs
    out["halfint"] = (rng.integers(-3, 4, (1, 2, H, W)) * 0.5).astype(np.float32)
    # left half pushes pixels out of the frame at once, right half pulls back in (sticky invalid)
    k = np.zeros((1, 2, H, W), np.float32)
    k[0, 0, :, : W // 2] = -(W // 2 + 2.25)
    k[0, 0, :, W // 2:] = +1.75
    k[0, 1] = rng.uniform(-0.5, 0.5, (H, W))
    out["exit"] = k
    return out


def splat_flows(N, H, W, rng):
    out = {}
    out["zero"] = np.zeros((N, 2, H, W), np.float32)
    out["integer"] = rng.integers(-4, 5, (N, 2, H, W)).astype(np.float32)
    out["half"] = (rng.integers(0, 2, (N, 2, H, W)) - 0.5).astype(np.float32)
    out["random3"] = rng.uniform(-3, 3, (N, 2, H, W)).astype(np.float32)
    out["oob"] = np.full((N, 2, H, W), max(H, W) + 1, np.float32)
    out["huge"] = rng.choice(np.array([-1e6, 1e6, 0.25], np.float32), (N, 2, H, W)).astype(np.float32)
    # converging: everything lands in a few pixels (heavy collisions)
    y, x = np.meshgrid(np.arange(H, dtype=np.float32), np.arange(W, dtype=np.float32), indexing="ij")
    cv = np.stack([(W / 2 - x) * 0.9 + 0.3, (H / 2 - y) * 0.9 - 0.2])[None].repeat(N, 0)
    out["converge"] = cv.astype(np.float32)
    return out


def main():
    ss, eim = load_reference()
    os.makedirs(OUT, exist_ok=True)       # files of other generators (make_golden_lz4.py, make_golden_nets.py) stay
    rng = np.random.default_rng(20260928)

    # ---- E1: euler_integration ------------------------------------------------------
    e1 = {}
    idx = 0
    for (H, W) in [(16, 24), (33, 47)]:
        for name, m in motion_fields(H, W, rng).items():
            for n in (0, 1, 2, 5, 17, 60):
                d, v = eim.euler_integration(torch.from_numpy(m), n)
                e1[f"c{idx}_motion"] = m
                e1[f"c{idx}_n"] = np.int32(n)
                e1[f"c{idx}_disp"] = d.numpy().astype(np.float32)
                e1[f"c{idx}_vis"] = v.numpy().astype(np.float32)
                e1[f"c{idx}_tag"] = np.array(f"{name}_{H}x{W}_n{n}")
                idx += 1
    # EulerIntegration module (batch wrapper, :58-71), B=3 with per-sample step counts
    mB = np.concatenate([motion_fields(16, 24, rng)[k] for k in ("smooth", "random3", "halfint")], 0)
    dest = torch.tensor([3, 0, 7])
    dB, vB = eim.EulerIntegration()(torch.from_numpy(mB), dest, show_visible_pixels=True)
    e1["module_motion"], e1["module_dest"] = mB, dest.numpy().astype(np.int32)
    e1["module_disp"], e1["module_vis"] = dB.numpy(), vB.numpy()
    e1["count"] = np.int32(idx)
    np.savez_compressed(os.path.join(OUT, "euler.npz"), **e1)

    # ---- S1: summation splat forward + backward ------------------------------------
    s1 = {}
    idx = 0
    for (N, C, H, W) in [(1, 1, 12, 20), (1, 3, 12, 20), (2, 4, 23, 37)]:
        for name, fl in splat_flows(N, H, W, rng).items():
            x = rng.standard_normal((N, C, H, W)).astype(np.float32)
            go = rng.standard_normal((N, C, H, W)).astype(np.float32)
            tx, tf = cudalike(x).requires_grad_(True), cudalike(fl).requires_grad_(True)
            out = ss._FunctionSoftsplat.apply(tx, tf)
            out.backward(cudalike(go))
            s1[f"c{idx}_in"], s1[f"c{idx}_flow"], s1[f"c{idx}_gout"] = x, fl, go
            s1[f"c{idx}_out"] = out.detach().numpy().astype(np.float32)
            s1[f"c{idx}_gin"] = tx.grad.numpy().astype(np.float32)
            s1[f"c{idx}_gflow"] = tf.grad.numpy().astype(np.float32)
            s1[f"c{idx}_tag"] = np.array(f"{name}_{N}x{C}x{H}x{W}")
            idx += 1
    s1["count"] = np.int32(idx)
    np.savez_compressed(os.path.join(OUT, "splat_sum.npz"), **s1)

    # ---- S2: FunctionSoftsplat, four modes (incl. zero-normaliser branch) -----------
    s2 = {}
    idx = 0
    for (N, C, H, W) in [(1, 3, 12, 20), (2, 3, 23, 37)]:
        flows = splat_flows(N, H, W, rng)
        for fname in ("random3", "integer", "oob", "converge"):
            for mode in ("summation", "average", "linear", "softmax"):
                x = rng.standard_normal((N, C, H, W)).astype(np.float32)
                met = rng.standard_normal((N, 1, H, W)).astype(np.float32)
                if mode == "linear":
                    met = np.abs(met) + 0.1
                out = ss.FunctionSoftsplat(cudalike(x), cudalike(flows[fname]), cudalike(met), mode)
                s2[f"c{idx}_in"], s2[f"c{idx}_flow"], s2[f"c{idx}_metric"] = x, flows[fname], met
                s2[f"c{idx}_mode"] = np.array(mode)
                s2[f"c{idx}_out"] = out.detach().numpy().astype(np.float32)
                s2[f"c{idx}_tag"] = np.array(f"{mode}_{fname}_{N}x{C}x{H}x{W}")
                idx += 1
    # ModuleSoftsplat keyword call as the models issue it (animating_softmax_splating.py:884-887)
    x = rng.standard_normal((1, 5, 12, 20)).astype(np.float32)
    fl = splat_flows(1, 12, 20, rng)["random3"]
    out = ss.ModuleSoftsplat("summation")(tenInput=cudalike(x), tenFlow=cudalike(fl),
                                          tenMetric=cudalike(np.ones((1, 1, 12, 20), np.float32)))
    s2["module_in"], s2["module_flow"], s2["module_out"] = x, fl, out.numpy().astype(np.float32)
    s2["count"] = np.int32(idx)
    np.savez_compressed(os.path.join(OUT, "splat_modes.npz"), **s2)

    # ---- M1: maximum splat family ---------------------------------------------------
    m1 = {}
    idx = 0
    for (N, C, H, W) in [(1, 1, 12, 20), (2, 2, 23, 37)]:
        for fname, fl in splat_flows(N, H, W, rng).items():
            x = rng.standard_normal((N, C, H, W)).astype(np.float32) * 3
            mx = ss._FunctionMaximumsplat.apply(cudalike(x), cudalike(fl))
            wn = ss.ModuleMaximumWarpNormsplat()(cudalike(x), cudalike(fl))
            m1[f"c{idx}_in"], m1[f"c{idx}_flow"] = x, fl
            m1[f"c{idx}_max"] = mx.numpy().astype(np.float32)
            m1[f"c{idx}_warpnorm"] = wn.numpy().astype(np.float32)
            m1[f"c{idx}_tag"] = np.array(f"{fname}_{N}x{C}x{H}x{W}")
            idx += 1
    m1["count"] = np.int32(idx)
    np.savez_compressed(os.path.join(OUT, "splat_max.npz"), **m1)

    # ---- E2: gradient of euler_integration w.r.t. the motion field (torch autograd through the reference's loop,
    # euler_integration_manipulator.py:36-55; the joint-training path, animating_softmax_splating.py:515-580)
    e2 = {}
    rg = np.random.default_rng(31)
    idx = 0
    for (H, W) in ((16, 24), (33, 47)):
        fields = motion_fields(H, W, rg)
        for fname in ("smooth", "random3", "exit"):
            for n in (1, 5, 17):
                m = torch.from_numpy(fields[fname]).clone().requires_grad_(True)
                disp, vis = eim.euler_integration(m, n)
                gout = torch.from_numpy(rg.standard_normal((1, 2, H, W)).astype(np.float32))
                (gm,) = torch.autograd.grad(disp, m, gout)
                e2[f"c{idx}_motion"] = fields[fname]
                e2[f"c{idx}_n"] = np.int32(n)
                e2[f"c{idx}_gout"] = gout.numpy()
                e2[f"c{idx}_gmotion"] = gm.numpy().astype(np.float32)
                e2[f"c{idx}_tag"] = np.array(f"{fname}_{H}x{W}_n{n}")
                idx += 1
    e2["count"] = np.int32(idx)
    np.savez_compressed(os.path.join(OUT, "euler_grad.npz"), **e2)

    # ---- L1: full-size digests (config C3 grid 768x1280; C2 grid 256x480) ------------------
    # inputs are regenerated from seeds by the tests; only digests of the reference's outputs are stored
    l1 = {}
    for tag, (H, W, C, steps) in {"c3": (768, 1280, 3, 30), "c2": (256, 480, 4, 59)}.items():
        r2 = np.random.default_rng(1000 + H)
        y, x = np.meshgrid(np.arange(H, dtype=np.float32), np.arange(W, dtype=np.float32), indexing="ij")
        u = 1.5 * np.sin(2 * np.pi * (2 * x / W + y / H) + 0.7)
        v = 1.5 * np.cos(2 * np.pi * (x / W - 1.5 * y / H) + 2.1)
        msk = (x >= 0.35 * W).astype(np.float32)
        motion = np.stack([u * msk, v * msk])[None].astype(np.float32)
        disp, vis = eim.euler_integration(torch.from_numpy(motion), steps)
        disp = disp.numpy().astype(np.float32)
        inp = r2.standard_normal((1, C, H, W)).astype(np.float32)
        out = ss._FunctionSoftsplat.apply(cudalike(inp), cudalike(disp)).numpy().astype(np.float32)
        pos = r2.integers(0, out.size, 4096)
        l1[f"{tag}_shape"] = np.array([1, C, H, W], np.int32)
        l1[f"{tag}_steps"] = np.int32(steps)
        l1[f"{tag}_disp_sum"] = disp.astype(np.float64).sum(axis=(2, 3))
        l1[f"{tag}_vis_sum"] = np.float64(vis.numpy().sum())
        l1[f"{tag}_disp_pos"] = r2.integers(0, disp.size, 4096)
        l1[f"{tag}_disp_val"] = disp.ravel()[l1[f"{tag}_disp_pos"]]
        l1[f"{tag}_out_sum"] = out.astype(np.float64).sum(axis=(2, 3))
        l1[f"{tag}_out_l2"] = np.sqrt((out.astype(np.float64) ** 2).sum(axis=(2, 3)))
        l1[f"{tag}_out_pos"] = pos
        l1[f"{tag}_out_val"] = out.ravel()[pos]
        l1[f"{tag}_holes"] = np.int64((out == 0).sum())
    np.savez_compressed(os.path.join(OUT, "large_digests.npz"), **l1)

    if True:
        from make_golden_pipeline import capture_pipeline, capture_pipeline_large, capture_v1_surface
        capture_pipeline(ss, eim, OUT, rng, cudalike)
        if "--skip-large" not in sys.argv:
            capture_pipeline_large(ss, OUT, cudalike)
        capture_v1_surface(ss, OUT, cudalike)

    total = sum(os.path.getsize(os.path.join(OUT, f)) for f in os.listdir(OUT))
    print("wrote", sorted(os.listdir(OUT)), f"{total / 1024:.0f} kB")


if __name__ == "__main__":
    try:
        sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
        main()
    finally:
        shutil.rmtree(TMP, ignore_errors=True)
