import sys, numpy as np, torch
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
import slr_sfs_amd as S
from oracle import oracle
oracle.build()
from test_gpu_frontends import smooth_motion
L=S._lib.lib()
H,W,C=256,480,64
rng=np.random.default_rng(2)
x=rng.standard_normal((1,C,H,W)).astype(np.float32); met=rng.standard_normal((1,1,H,W)).astype(np.float32)
flow=oracle.euler_integration(smooth_motion(H,W),30)[0]
for fe in (2,):
    L.slr_splat_set_front_end(fe)
    for mode,m in (("summation",None),("softmax",met)):
        out=S.FunctionSoftsplat(torch.from_numpy(x).cuda(),torch.from_numpy(flow).cuda(),None if m is None else torch.from_numpy(m).cuda(),mode).cpu().numpy()
        ref=oracle.function_softsplat(x,flow,m,mode)
        d=np.abs(out-ref).max(axis=1)[0]
        bad=np.argwhere(d>1e-4)
        print(fe,mode,"max err",d.max(),"bad px",len(bad))
        if len(bad):
            ty=bad[:,0]//8; tx=bad[:,1]//64
            tiles=sorted(set(zip(ty.tolist(),tx.tolist())))
            print(" bad tiles",tiles[:10], "rows in tile", sorted(set((bad[:,0]%8).tolist())))
