"""Phase timeline of the FUSED tile kernel (two directions) -- needs the -DSLR_TRACE build."""
import ctypes, os, sys
import numpy as np, torch
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
os.environ["SLR_SFS_AMD_LIB"] = os.path.join(R, "slr-sfs_amd/lib/var_trace.so")
import slr_sfs_amd as S
from bench import smooth_motion, H, W, NFRAMES
L = S._lib.lib()
L.slr_debug_trace.argtypes = [ctypes.c_void_p]
dev = torch.device("cuda:0")
fs = torch.randn(1, 64, H, W, device=dev); Z = torch.randn(1, 1, H, W, device=dev)
cs = S.synthesis.ClipSynthesizer(fs, Z, torch.from_numpy(smooth_motion(H, W)).to(dev), NFRAMES)
for t in (1, 30):
    cs.features(t)
    nb = 4096
    buf = torch.zeros(nb * 40 * 2, dtype=torch.int64, device=dev)
    L.slr_debug_trace(buf.data_ptr())
    cs.features(t)
    torch.cuda.synchronize()
    L.slr_debug_trace(None)
    tr = buf.cpu().numpy().reshape(-1, 40)[:nb]
    tr = tr[tr[:, 3] > 0]
    nch = 16
    last = np.where(tr[:, 4:31] > 0, tr[:, 4:31], 0).max(axis=1)
    tot = last - tr[:, 0]
    print(f"t={t}: blocks {len(tr)}  total cycles p50/p90/p99/max {np.percentile(tot,[50,90,99,100])}")
    print("   phase1a", np.median(tr[:,1]-tr[:,0]), "scan+rec", np.median(tr[:,3]-tr[:,2]), " first stage wait", np.median(tr[:,4]-tr[:,3]))
    g = [np.median(tr[:, 6+3*c]-tr[:, 5+3*c]) for c in range(8)]
    b = [np.median(tr[:, 5+3*c]-tr[:, 4+3*c]) for c in range(8)]
    st = [np.median(tr[:, 7+3*c]-tr[:, 6+3*c]) for c in range(8)]
    print("   gather+store per chunk", g, "\n   barrier", b, "\n   barrier+stage", st)
    print("   max rec/pixel in slow blocks:", tr[np.argsort(tot)[-5:], 33], " corr(total,maxrec)", np.corrcoef(tot, tr[:,33])[0,1])
