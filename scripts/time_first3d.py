"""time the 3-D first-layer kernels at cfg5's shape (2 x 160^3, bf16): python scripts/time_first3d.py"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mdeical_image_segmentation_amd import ops
N, S, Co, Cp = 2, 160, 32, 64
dev = "cuda"
g = torch.Generator(device=dev).manual_seed(0)
x = torch.randn(N, S, S, S, device=dev, generator=g)
w = torch.randn(Co, 1, 3, 3, 3, device=dev, generator=g) * 0.2
scale, shift = torch.ones(N, 4, device=dev), torch.zeros(N, 4, device=dev)
mean, rstd = torch.zeros(N, 1, device=dev), torch.ones(N, 1, device=dev)
gamma, beta = torch.ones(1, device=dev), torch.zeros(1, device=dev)
for dt in (torch.bfloat16, torch.float32):
    y = torch.empty(N, S, S, S, Cp, dtype=dt, device=dev)
    gd = torch.randn(N, S, S, S, Cp, device=dev, generator=g).to(dt)
    dw = torch.zeros(Co, 1, 3, 3, 3, device=dev); dg = torch.zeros(1, device=dev); db = torch.zeros(1, device=dev)
    def t(f, n=5):
        f(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n): f()
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) / n
    res = {}
    for sw in (0, 1):
        ops.dispatch_override("MIS_FIRST3D_NOMFMA", sw)
        res[sw] = (t(lambda: ops.first3d_fwd(x, scale, shift, 4, w, Co, y, Cp)), t(lambda: ops.first3d_bwd(x, mean, rstd, gamma, beta, gd, Cp, w, Co, dw, dg, db)))
    ops.dispatch_override("MIS_FIRST3D_NOMFMA", -1)
    print(dt, "fwd mfma %.3f ms vs tiled %.3f ms | bwd mfma %.3f ms vs tiled %.3f ms" % (res[0][0], res[1][0], res[0][1], res[1][1]))
