"""HBM bandwidth reference points on the box (GPU): torch copy / add on 1 GiB bf16 tensors, next to the engine's streaming kernels.   python scripts/bw_probe.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mdeical_image_segmentation_amd import ops  # noqa: E402


def timeit(f, n=10):
    f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        f()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


N, H, W, C = 32, 512, 512, 64
x = torch.randn(N, H, W, C, device="cuda").to(torch.bfloat16)
y = torch.empty_like(x)
z = torch.empty_like(x)
gb = x.numel() * 2 / 1e9
t = timeit(lambda: y.copy_(x))
print(f"torch copy      {t:.3f} ms  {2 * gb / t:.2f} TB/s")
t = timeit(lambda: torch.add(x, y, out=z))
print(f"torch add       {t:.3f} ms  {3 * gb / t:.2f} TB/s")
t = timeit(lambda: x.float().sum() if False else torch.relu_(y))
print(f"torch relu_     {t:.3f} ms  {2 * gb / t:.2f} TB/s")
pooled = torch.empty(N, H // 2, W // 2, C, device="cuda", dtype=torch.bfloat16)
pb = torch.empty(N, H // 2, W // 2, C, device="cuda", dtype=torch.uint8)
t = timeit(lambda: ops.maxpool2_fwd(x, pooled, pbits=pb))
print(f"maxpool_fwd_pb  {t:.3f} ms  {(gb + gb / 4 + gb / 8) / t:.2f} TB/s")
g = torch.randn(N, H // 2, W // 2, C, device="cuda").to(torch.bfloat16)
t = timeit(lambda: ops.maxpool2_bwd(None, g, z, add=y, relu_mask=True, pbits=pb))
print(f"maxpool_bwd_pb  {t:.3f} ms  {(2 * gb + gb / 4 + gb / 8) / t:.2f} TB/s")
t = timeit(lambda: ops.maxpool2_bwd(x, g, z, add=y, relu_mask=True))
print(f"maxpool_bwd     {t:.3f} ms  {(3 * gb + gb / 4) / t:.2f} TB/s")
