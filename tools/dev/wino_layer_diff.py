"""Every 3x3 layer call of a frame on the fp32 rung, run as Winograd AND direct on the same inputs: where do they differ?  python tools/dev/wino_layer_diff.py"""
import os, sys, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_large_golden as T
from slr_sfs_amd import nets
gd = os.path.join(ROOT, "tests", "golden")
g = np.load(f"{gd}/native_frames_768.npz")
S, N = int(g["S"]), int(g["N"])
img, motion, _ = T.NF.e2e_inputs(S, N)
img, motion = torch.from_numpy(img).cuda(), torch.from_numpy(motion).cuda()
base = T._baseline(gd).cuda(); base.convs = "fp32"
log = []
def unb8(t, lay_out):
    if not (lay_out & nets.OUT_B8): return t
    n, c, h, w = t.shape
    return t.view(n, c // 8, h, w, 8).permute(0, 1, 4, 2, 3).reshape(n, c, h, w)
orig_pf, orig_cv = nets.PartialConv.forward, nets.Conv.conv
def pf(self, x, mask, residual=None, next_bn=None, pre_bn=None, layout=0):
    if self.k != 3 or not x.is_cuda: return orig_pf(self, x, mask, residual, next_bn, pre_bn, layout)
    with nets.fp32_kernels(winograd=False):
        od, ud = orig_pf(self, x, mask, residual, next_bn, pre_bn, layout)
    with nets.fp32_kernels(winograd=True):
        ow, uw = orig_pf(self, x, mask, residual, next_bn, pre_bn, layout)
    d = unb8((ow - od).abs(), layout)
    i = int(d.argmax()); c, y, xx = np.unravel_index(i, d.shape[1:])
    frac = float(((mask > 0) & (mask < 1)).float().mean()) if mask is not None else -1
    log.append(("pconv", tuple(x.shape), self.weight.shape[0], float(d.max()), float(od.abs().max()), (int(c), int(y), int(xx)), float((uw - ud).abs().max()), frac,
                "derived" if mask is None else "mask", "pre" if pre_bn is not None else "-", "next" if next_bn is not None else "-", "res" if residual is not None else "-"))
    return od, ud
def cv(self, x, bias, pre_bn=None, residual=None, layout=0):
    if self.k != 3 or not x.is_cuda or self.weight.shape[0] <= 4: return orig_cv(self, x, bias, pre_bn, residual, layout)
    with nets.fp32_kernels(winograd=False):
        od = orig_cv(self, x, bias, pre_bn, residual, layout)
    with nets.fp32_kernels(winograd=True):
        ow = orig_cv(self, x, bias, pre_bn, residual, layout)
    d = unb8((ow - od).abs(), layout)
    i = int(d.argmax()); c, y, xx = np.unravel_index(i, d.shape[1:])
    log.append(("conv", tuple(x.shape), self.weight.shape[0], float(d.max()), float(od.abs().max()), (int(c), int(y), int(xx)), 0.0, -1, "-", "pre" if pre_bn is not None else "-", "-", "res" if residual is not None else "-"))
    return od
nets.PartialConv.forward, nets.Conv.conv = pf, cv
base.synthesize(img, motion, N, frames=[30])
for r in log:
    print(f"{r[0]:5s} in {str(r[1]):24s} -> {r[2]:4d}: max |wino - direct| {r[3]:.2e} (output range {r[4]:.2e}, rel {r[3] / max(r[4], 1e-30):.1e}) at {r[5]}, um diff {r[6]:.1e}, mask fractional {r[7]:.4f} {' '.join(r[8:])}")
