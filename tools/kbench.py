#!/usr/bin/env python
"""Kernel-level timing of the splat path on one MI355X (development aid; bench.py is the
contract).  Inputs are synthetic.  Times are HIP-event medians on
torch's current stream, which is the stream the library launches on."""
import argparse
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import slr_sfs_amd as S  # noqa: E402
from slr_sfs_amd._lib import check, lib, ptr, stream_of, workspace  # noqa: E402


def smooth_motion(H, W, seed=0, amp=1.5):
    rng = np.random.default_rng(seed)
    p1, p2 = rng.uniform(0, 2 * np.pi, 2)
    y, x = np.meshgrid(np.arange(H, dtype=np.float32), np.arange(W, dtype=np.float32), indexing="ij")
    u = amp * np.sin(2 * np.pi * (2 * x / W + y / H) + p1)
    v = amp * np.cos(2 * np.pi * (x / W - 1.5 * y / H) + p2)
    m = (x >= 0.35 * W).astype(np.float32)
    return torch.from_numpy(np.stack([u * m, v * m])[None].astype(np.float32)).cuda()


def timeit(fn, iters=30, warm=5):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(iters):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        fn()
        b.record()
        b.synchronize()
        ts.append(a.elapsed_time(b) * 1e3)
    ts.sort()
    return ts[len(ts) // 2], ts[0]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--H", type=int, default=768)
    ap.add_argument("--W", type=int, default=1280)
    ap.add_argument("--C", type=int, default=65)
    ap.add_argument("--iters", type=int, default=30)
    ap.add_argument("--quick", action="store_true")
    a = ap.parse_args()
    H, W, C = a.H, a.W, a.C
    g = torch.Generator(device="cuda").manual_seed(0)
    x = torch.randn(1, C, H, W, device="cuda", generator=g)
    out = torch.empty_like(x)
    m = smooth_motion(H, W)
    flows = {"identity": torch.zeros(1, 2, H, W, device="cuda")}
    dall, _ = S.euler_integration_all(m, 60)
    flows["euler_t30"] = dall[30:31].contiguous()
    flows["euler_t59"] = dall[59:60].contiguous()
    flows["incoherent"] = (torch.rand(1, 2, H, W, device="cuda", generator=g) * 16 - 8)
    B = (2 * C + 2) * H * W * 4
    L = lib()
    ws = workspace(x, "a", 1, C, H, W)
    st = stream_of(x)
    for name, fl in flows.items():
        def full():
            check(L.slr_softsplat_forward(ptr(x), ptr(fl), ptr(out), 1, C, H, W, ptr(ws), ws.numel(), 0, st), "f")

        def binonly():
            check(L.slr_splat_bin(ptr(fl), 1, H, W, ptr(ws), ws.numel(), st), "b")

        def splatonly():
            check(L.slr_softsplat_forward(ptr(x), ptr(fl), ptr(out), 1, C, H, W, ptr(ws), ws.numel(), 1, st), "s")
        tf = timeit(full, a.iters)
        tb = timeit(binonly, a.iters)
        binonly()
        tsp = timeit(splatonly, a.iters)
        print(json.dumps({"case": name, "shape": [C, H, W], "full_us": tf, "bin_us": tb, "splat_us": tsp,
                          "alg_MB": B / 1e6, "splat_TBps": B / tsp[0] / 1e6, "full_TBps": B / tf[0] / 1e6}))
    if a.quick:
        return
    # euler all-frames, both directions
    te = timeit(lambda: S.euler_integration_all(m, 60, want_visible=False), 10, 2)
    print(json.dumps({"case": "euler_all_60", "us": te}))
    te = timeit(lambda: S.euler_integration(m, 59), 10, 2)
    print(json.dumps({"case": "euler_59", "us": te}))
    # fused frame synthesis (64 features + norm, two directions)
    fs = torch.randn(1, 64, H, W, device="cuda", generator=g)
    Z = torch.randn(1, 1, H, W, device="cuda", generator=g)
    cs = S.synthesis.ClipSynthesizer(fs, Z, m, 60)
    for t in (1, 30, 59):
        tt = timeit(lambda: cs.features(t), a.iters)
        print(json.dumps({"case": f"synth_frame_t{t}", "us": tt, "ref_alg_MB": 2 * (2 * 65 + 2) * H * W * 4 / 1e6}))


if __name__ == "__main__":
    main()
