"""time mis_conv_igemm (3x3, bf16, bs 32, bias + ReLU) on the given layers: python scripts/bench_one_conv.py H Cin Cout [H Cin Cout ...]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mdeical_image_segmentation_amd import ops  # noqa: E402

MASK = "--mask" in sys.argv          # dgrad form: no bias / ReLU, ReLU mask of another tensor in the epilogue
v = [int(a) for a in sys.argv[1:] if a != "--mask"]
for H, Cin, Cout in zip(v[0::3], v[1::3], v[2::3]):
    g = torch.Generator(device="cuda").manual_seed(1)
    x = torch.randn(32, H, H, Cin, device="cuda", generator=g).to(torch.bfloat16)
    w = (torch.randn(9, Cout, Cin, device="cuda", generator=g) * (9 * Cin) ** -0.5).to(torch.bfloat16)
    b = torch.randn(Cout, device="cuda", generator=g)
    y = torch.empty(32, H, H, Cout, device="cuda", dtype=torch.bfloat16)
    m = torch.randn(32, H, H, Cout, device="cuda", generator=g).to(torch.bfloat16) if MASK else None
    kw = dict(mask=m) if MASK else dict(bias=b, relu=True)
    best = 1e9
    for r in range(4):
        ops.conv_igemm(x, w, y, ksize=3, Cin=Cin, Cout=Cout, **kw)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            ops.conv_igemm(x, w, y, ksize=3, Cin=Cin, Cout=Cout, **kw)
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / 5)
    print(f"{H:4d}^2 {Cin:5d}->{Cout:<5d} {best:7.3f} ms {2.0 * 32 * H * H * 9 * Cin * Cout / best / 1e9:6.0f} TF  {ops.conv_last_dispatch()}{' mask' if MASK else ''}", flush=True)
