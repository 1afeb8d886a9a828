import os, sys, torch
sys.path.insert(0, "tools"); sys.path.insert(0, ".")
sys.argv = ["x", "none"]
import frontend_bench as F
S = F.S; dev = F.dev
for C, h, w in ((65, 256, 256), (65, 384, 640), (65, 512, 512), (65, 512, 896), (65, 640, 1024)):
    for N in ((1, 2) if h == 256 else (1,)):
        x, met = torch.randn(N, C, h, w, device=dev), torch.randn(N, 1, h, w, device=dev)
        alg = N * (2 * C + 3) * h * w * 4
        mo = torch.from_numpy(F.smooth_motion(h, w)).to(dev)
        for name, fl in (("inc", torch.rand(1, 2, h, w, device=dev) * 16 - 8), ("t30", S.euler_integration(mo, 30)[0]), ("t59", S.euler_integration(mo, 59)[0])):
            fl = fl.expand(N, -1, -1, -1).contiguous()
            F.measure(f"{N}x{C}x{h}x{w} {name}", x, fl, met, "softmax", alg)
