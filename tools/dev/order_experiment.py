"""Experiment: how much would the scan front end gain at 768x1280 if its tiles were started heaviest first?
(library built with -DSLR_SCAN_ORDER_HOOK: the tile kernel takes its block -> tile map from a given array).
Orders tried: none (spatial), exact entry counts (computed here with torch), and a cheap estimate a box kernel could
produce itself: number of source tiles whose destination box touches the tile."""
import ctypes, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
os.environ["SLR_SFS_AMD_LIB"] = os.path.join(ROOT, "slr-sfs_amd/lib/var_order.so")
import slr_sfs_amd as S
from kbench import smooth_motion
sys.argv = sys.argv[:1]
from bench import _graph_call_us
L = S._lib.lib()
L.slr_debug_scan_order.argtypes = [ctypes.c_void_p]
L.slr_debug_scan_order.restype = None
L.slr_splat_set_scan_max_tiles(2 ** 31 - 1)
H, W, C = 768, 1280, 65
TH, TW = 8, 64
tx, ty = W // TW, H // TH
x = torch.randn(1, C, H, W, device="cuda")
m = smooth_motion(H, W)


def tile_counts(fl):
    yy, xx = torch.meshgrid(torch.arange(H, device="cuda"), torch.arange(W, device="cuda"), indexing="ij")
    X, Y = xx + fl[0, 0], yy + fl[0, 1]
    x0, y0 = torch.floor(X).long(), torch.floor(Y).long()
    cnt = torch.zeros(tx * ty, dtype=torch.long, device="cuda")
    seen = None
    keys = []
    for dx in (0, 1):
        for dy in (0, 1):
            cx, cy = x0 + dx, y0 + dy
            ok = (cx >= 0) & (cx < W) & (cy >= 0) & (cy < H)
            t = (cy.clamp(0, H - 1) // TH) * tx + cx.clamp(0, W - 1) // TW
            keys.append(torch.where(ok, t, torch.full_like(t, -1)))
    k = torch.stack(keys, 0).view(4, -1)                       # tiles touched per pixel (with duplicates)
    k, _ = torch.sort(k, 0)
    dup = torch.zeros_like(k, dtype=torch.bool)
    dup[1:] = k[1:] == k[:-1]
    k = torch.where(dup, torch.full_like(k, -1), k)
    k = k[k >= 0]
    cnt.scatter_add_(0, k, torch.ones_like(k))
    return cnt


def box_estimate(fl):
    """per output tile: number of source tiles whose box of NW corners touches it (what the tile kernel calls candidates)"""
    yy, xx = torch.meshgrid(torch.arange(H, device="cuda"), torch.arange(W, device="cuda"), indexing="ij")
    x0, y0 = torch.floor(xx + fl[0, 0]).long(), torch.floor(yy + fl[0, 1]).long()
    v = lambda t: t.view(ty, TH, tx, TW).permute(0, 2, 1, 3).reshape(ty * tx, -1)
    bx0, bx1, by0, by1 = v(x0).min(1)[0], v(x0).max(1)[0], v(y0).min(1)[0], v(y0).max(1)[0]
    est = torch.zeros(tx * ty, dtype=torch.long, device="cuda")
    for s in range(tx * ty):
        tx0, tx1 = max(0, int(bx0[s]) // TW), min(tx - 1, (int(bx1[s]) + 1) // TW)
        ty0_, ty1 = max(0, int(by0[s]) // TH), min(ty - 1, (int(by1[s]) + 1) // TH)
        if tx0 <= tx1 and ty0_ <= ty1:
            for r in range(ty0_, ty1 + 1):
                est[r * tx + tx0:r * tx + tx1 + 1] += 1
    return est


for name, steps in (("t30", 30), ("t59", 59)):
    fl = S.euler_integration(m, steps)[0]
    f = lambda: S.FunctionSoftsplat(x, fl, None, "summation")
    cnt = tile_counts(fl)
    est = box_estimate(fl)
    res = []
    for tag, order in (("spatial", None), ("exact counts", torch.argsort(cnt, descending=True)),
                       ("box estimate", torch.argsort(est, descending=True, stable=True))):
        keep = None
        if order is not None:
            keep = order.to(torch.int32).contiguous()
            L.slr_debug_scan_order(ctypes.c_void_p(keep.data_ptr()))
        else:
            L.slr_debug_scan_order(None)
        out = f()
        res.append(f"{tag} {_graph_call_us(f):6.1f} us")
    L.slr_debug_scan_order(None)
    print(name, "max entries", int(cnt.max()), "tiles > 1024:", int((cnt > 1024).sum()), "corr(est, cnt)",
          float(np.corrcoef(est.cpu().numpy(), cnt.cpu().numpy())[0, 1]), "|", " | ".join(res))
