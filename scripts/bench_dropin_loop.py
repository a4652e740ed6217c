"""The drop-in nn.Module path without the HF Trainer around it: UNetModel (autograd.Function over the fused engine) + clip_grad_norm_ + torch AdamW on a
GPU-resident batch.   MISAMD_DTYPE=bf16 python scripts/bench_dropin_loop.py"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mdeical_image_segmentation_amd.dropin as d  # noqa: E402

d.install()
from unet2d import UNetConfig, UNetModel  # noqa: E402

B, S = 32, 512
torch.manual_seed(0)
m = UNetModel(UNetConfig(in_channels=1, out_channels=2, unet_type="UNet")).cuda().train()
opt = torch.optim.AdamW(m.parameters(), lr=1e-4, weight_decay=1e-3, fused=True)
x = torch.randn(B, 1, S, S, device="cuda")
y = torch.randint(0, 2, (B, S, S), device="cuda")


def step(parts):
    t0 = time.perf_counter()
    opt.zero_grad(set_to_none=True)
    out = m(images=x, labels=y)
    torch.cuda.synchronize(); t1 = time.perf_counter()
    out.loss.backward()
    torch.cuda.synchronize(); t2 = time.perf_counter()
    torch.nn.utils.clip_grad_norm_(m.parameters(), 1.0)
    opt.step()
    torch.cuda.synchronize(); t3 = time.perf_counter()
    parts[0] += t1 - t0; parts[1] += t2 - t1; parts[2] += t3 - t2


for _ in range(3):
    step([0, 0, 0])
parts = [0.0, 0.0, 0.0]
n = 10
for _ in range(n):
    step(parts)
tot = sum(parts) / n
print(f"drop-in module loop {os.environ.get('MISAMD_DTYPE', 'f32')} bs={B}: {tot * 1e3:.1f} ms/step = {B / tot:.1f} images/s "
      f"(forward {parts[0] / n * 1e3:.1f}, backward {parts[1] / n * 1e3:.1f}, clip+AdamW {parts[2] / n * 1e3:.1f} ms)")
