#!/usr/bin/env python
"""The front ends of the one-flow operator (bins | scan | rows) on the small grids of config C2 and at 768x1280:
tile kernel alone (events recorded by the library around that launch) and the whole call.
`graph`: the call captured into a HIP graph of 20 calls and replayed -- GPU time per call without the host's launch pace."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import slr_sfs_amd as S
from slr_sfs_amd import synthesis
from bench import smooth_motion
from kbench import timeit
dev = torch.device("cuda:0")
L = S._lib.lib()
which = (sys.argv[1:] or ["small", "full"]) if __name__ == "__main__" else []


def graph_us(fn, reps=20, iters=10):
    fn(); torch.cuda.synchronize()
    st = torch.cuda.Stream()
    with torch.cuda.stream(st):
        fn()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=st):
            for _ in range(reps):
                fn()
        g.replay(); torch.cuda.synchronize()
        ts = []
        for _ in range(iters):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(); g.replay(); b.record(); b.synchronize()
            ts.append(a.elapsed_time(b) * 1e3 / reps)
    ts.sort()
    return ts[len(ts) // 2]


def measure(tag, x, fl, met, mode, alg):
    row = []
    for fe, code in (("scan", 1), ("rows", 2)):
        prev = L.slr_splat_set_front_end(code)
        f = lambda: S.FunctionSoftsplat(x, fl, met, mode)
        synthesis.kernel_timing = []
        for _ in range(25):
            synthesis._arm_timer(x)
            f()
        torch.cuda.synchronize()
        us = sorted(a.elapsed_time(b) * 1e3 for a, b, _ in synthesis.kernel_timing[5:])
        synthesis.kernel_timing = None
        k = sum(us) / len(us)
        call = timeit(f, 30)[0]
        try:
            gus = graph_us(f)
        except Exception as e:                                   # noqa
            gus = float("nan")
        row.append(f"{fe}: tile {k:6.1f} call {call:6.1f} graph {gus:6.1f} us ({alg / gus / 1e3 / 8000:.3f})")
        L.slr_splat_set_front_end(prev)
    print(f"{tag:22s} " + " | ".join(row), flush=True)


if "small" in which:
    for C, h, w in ((64, 256, 480), (64, 128, 240), (65, 384, 640)):
        x, met = torch.randn(1, C, h, w, device=dev), torch.randn(1, 1, h, w, device=dev)
        alg = (2 * C + 3) * h * w * 4
        measure(f"{C}x{h}x{w} inc softmax", x, torch.rand(1, 2, h, w, device=dev) * 16 - 8, met, "softmax", alg)
        mo = torch.from_numpy(smooth_motion(h, w)).to(dev)
        measure(f"{C}x{h}x{w} t30 softmax", x, S.euler_integration(mo, 30)[0], met, "softmax", alg)
if "full" in which:
    H, W = 768, 1280
    motion = torch.from_numpy(smooth_motion(H, W)).to(dev)
    x = torch.randn(1, 65, H, W, device=dev)
    alg = (2 * 65 + 2) * H * W * 4
    flows = {"id": torch.zeros(1, 2, H, W, device=dev), "t30": S.euler_integration(motion, 30)[0],
             "t59": S.euler_integration(motion, 59)[0], "inc": torch.rand(1, 2, H, W, device=dev) * 16 - 8}
    for name, fl in flows.items():
        measure(f"65x768x1280 {name} sum", x, fl, None, "summation", alg)
