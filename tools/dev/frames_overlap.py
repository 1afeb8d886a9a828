#!/usr/bin/env python
"""The measurement behind DESIGN.md 3.2.4: do consecutive frames of a clip lose time by waiting for each other?
The same 60 frames (fused two-direction splat, 64 features, 768x1280) issued one frame per launch (a) on one stream,
(b) alternately on 2 / 3 / 4 streams, (c) 8 frames per launch (slr_synth_group_clip_batch).  Round 2, one MI355X:
(a) 243 us per frame, (b) 190-200, (c) 195-205 (block ranges end to end) / 178-186 (block groups interleaved)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))); sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tools")); sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import slr_sfs_amd as S
from bench import smooth_motion, H, W, NFRAMES
dev = torch.device("cuda:0")
fs = torch.randn(1, 64, H, W, device=dev)
Z = torch.randn(1, 1, H, W, device=dev)
motion = torch.from_numpy(smooth_motion(H, W)).to(dev)
cs = S.synthesis.ClipSynthesizer(fs, Z, motion, NFRAMES)
outs = torch.empty(8, 64, H, W, device=dev)
REP = 3


def timed(fn):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(REP):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / (REP * NFRAMES) * 1e6


def one_stream():
    for t in range(NFRAMES):
        cs.features(t, out=outs[t % 8:t % 8 + 1])


def streams(k):
    ss = [torch.cuda.Stream() for _ in range(k)]

    def run():
        for s in ss:
            s.wait_stream(torch.cuda.current_stream())
        for t in range(NFRAMES):
            with torch.cuda.stream(ss[t % k]):
                cs.features(t, out=outs[t % 8:t % 8 + 1])
        for s in ss:
            torch.cuda.current_stream().wait_stream(s)
    return run


def batched():
    B = S.synthesis.MAX_BATCH
    for t0 in range(0, NFRAMES, B):
        ts = list(range(t0, min(t0 + B, NFRAMES)))
        cs.features_batch(ts, outs[:len(ts)])


print(f"one frame per launch, one stream : {timed(one_stream):6.1f} us per frame")
for k in (2, 3, 4):
    print(f"one frame per launch, {k} streams : {timed(streams(k)):6.1f} us per frame")
print(f"{S.synthesis.MAX_BATCH} frames per launch              : {timed(batched):6.1f} us per frame")
