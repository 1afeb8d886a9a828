"""The clip kernels on plane-blocked values (SLR_SYNTH_VALUES_B4) against the planar tensor, and slr_pack_planes4.  python tools/dev/b4_check.py"""
import os, sys, torch
sys.path.insert(0, "/root/repo")
import slr_sfs_amd as S
from slr_sfs_amd import synthesis
from bench import smooth_motion, H, W, NFRAMES
dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(0)
fs = torch.randn(1, 64, H, W, device=dev, generator=g); Z = torch.randn(1, 1, H, W, device=dev, generator=g)
motion = torch.from_numpy(smooth_motion(H, W)).to(dev)
outs = {}
for b4 in (True, False, 'again'):
    synthesis.USE_B4 = b4 is True
    cs = synthesis.ClipSynthesizer(fs, Z, motion, NFRAMES)
    assert (cs.fs4 is not None) == (b4 is True)
    ts = list(range(22, 38))
    out = torch.empty(len(ts), 64, H, W, device=dev)
    cs.features_batch(ts, out)
    outs[b4] = out
torch.cuda.synchronize()
if "--out-b8" in sys.argv:                         # experiment build (-DSLR_OUT_B8_EXP=1): the B4 kernels write [C/8][H][W][8]
    o = outs[True]
    outs[True] = o.view(o.shape[0], 8, H, W, 8).permute(0, 1, 4, 2, 3).reshape(o.shape)
print("B4 vs planar: equal", torch.equal(outs[True], outs[False]), float((outs[True] - outs[False]).abs().max()), "| planar vs planar again: equal", torch.equal(outs[False], outs["again"]), float((outs[False] - outs["again"]).abs().max()))
p = synthesis.pack_planes4(fs)
ref = fs.view(1, 16, 4, H, W).permute(0, 1, 3, 4, 2).contiguous().view_as(fs)
print("pack ok", torch.equal(p, ref))
