#!/usr/bin/env python
"""Small grids of the one-flow operator (scan front end): config C2 as stated and with smooth Euler flows, 128x240, 384x640, and the
reference's TRAINING shape [2,65,256,256] (forward + backward) -- GPU time of everything a call launches (20 calls in a HIP graph), and
every case checked against the CPU oracle.  Development aid (SLR_SFS_AMD_LIB=variant); bench.py's roofline_dropin carries the same legs.
    python tools/small_grid_bench.py [--no-check]"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import slr_sfs_amd as S
from bench import smooth_motion, _graph_call_us
from slr_sfs_amd._lib import check, lib, ptr, stream_of
dev = torch.device("cuda:0")
CHECK = "--no-check" not in sys.argv
if CHECK:
    from oracle import oracle as o
    o.build()
L = lib()
if os.environ.get("SINK_SHAPE"):                                     # "<pieces>,<groups>" of the sink launch (slr_splat_set_scan_shape)
    sp, sg = (int(v) for v in os.environ["SINK_SHAPE"].split(","))
    L.slr_splat_set_scan_shape(0, 0, sp, sg)
torch.manual_seed(0)


def flow_of(h, w, kind, n=1):
    if kind == "incoherent":
        return torch.rand(n, 2, h, w, device=dev) * 16 - 8
    t = int(kind[1:])
    mo = torch.from_numpy(np.concatenate([smooth_motion(h, w, seed=i) for i in range(n)], 0)).to(dev)
    return S.EulerIntegration()(mo, torch.full((n,), t, device=dev)).contiguous()


print(os.path.basename(os.environ.get("SLR_SFS_AMD_LIB", "default")))
for (n, c, h, w, mode) in ((1, 64, 256, 480, "softmax"), (1, 64, 128, 240, "softmax"), (1, 65, 384, 640, "softmax"), (2, 65, 256, 256, "summation")):
    x, met = torch.randn(n, c, h, w, device=dev), torch.randn(n, 1, h, w, device=dev)
    for kind in ("incoherent", "t30", "t59"):
        fl = flow_of(h, w, kind, n)
        f = (lambda: S.FunctionSoftsplat(x, fl, met, mode)) if mode != "summation" else (lambda: S.softsplat._FunctionSoftsplat.apply(x, fl))
        with torch.no_grad():
            us = _graph_call_us(f)
            out = f()
        alg = ((2 * c + 3) if mode != "summation" else (2 * c + 2)) * n * h * w * 4
        line = f"{n}x{c}x{h}x{w} {mode:9s} {kind:10s} fwd {us:7.1f} us ({alg / us / 1e3 / 8000:.3f})"
        if mode == "summation":
            go, gi, gf = torch.randn_like(x), torch.empty_like(x), torch.empty(n, 2, h, w, device=dev)
            nb = int(L.slr_softsplat_backward_ws_bytes(n, c, h, w))
            bws = torch.empty(max(nb, 1), dtype=torch.uint8, device=dev)
            bus = _graph_call_us(lambda: check(L.slr_softsplat_backward_ws(ptr(x), ptr(fl), ptr(go), ptr(gi), ptr(gf), n, c, h, w, ptr(bws), nb, stream_of(x)), "bwd"))
            line += f"  bwd {bus:6.1f} us ({(3 * c + 4) * n * h * w * 4 / bus / 1e3 / 8000:.3f})"
        if CHECK:
            ref = o.function_softsplat(x.cpu().numpy(), fl.cpu().numpy(), met.cpu().numpy(), mode) if mode != "summation" else \
                o.softsplat_forward(x.cpu().numpy(), fl.cpu().numpy())
            err = np.abs(out.cpu().numpy() - ref)
            line += f"  max|err| {err.max():.2e} (scale {np.abs(ref).max():.1f})"
        print(line, flush=True)
