"""Timing of the UNet 3+ up-branch gather kernels (csrc/upconv.hip) alone: python scripts/bench_upconv.py [batch]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mdeical_image_segmentation_amd import ops  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 16
dt = torch.bfloat16 if os.environ.get("MISAMD_DTYPE", "bf16") == "bf16" else torch.float32
C = 64
for s in (2, 4, 8, 16):
    h = 512 // s
    z = torch.randn(N, h, h, 9 * C, device="cuda").to(dt)
    y = torch.empty(N, 512, 512, C, device="cuda", dtype=dt)
    dy = torch.randn(N, 512, 512, C, device="cuda").to(dt)
    dz = torch.empty_like(z)
    b = torch.randn(C, device="cuda")
    res = []
    for fn in (lambda: ops.upconv_gather_fwd(z, y, s, C, bias=b), lambda: ops.upconv_gather_bwd(dy, dz, s, C)):
        for _ in range(2):
            fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            fn()
        e1.record()
        torch.cuda.synchronize()
        res.append(e0.elapsed_time(e1) / 5)
    print(f"s={s:2d} N={N} 512x512x{C} {str(dt)[6:]}: fwd {res[0]:.3f} ms  bwd {res[1]:.3f} ms   (y: {y.numel() * y.element_size() / 1e9:.2f} GB)")
