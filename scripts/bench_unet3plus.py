"""Timing of the per-layer UNet 3+ path (not the headline benchmark): UNetModel(unet_type='UNet_3Plus'), 3->1 channels, 512x512,
forward + SegmentationLoss + backward + torch AdamW.   MISAMD_DTYPE=bf16 python scripts/bench_unet3plus.py [batch]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mdeical_image_segmentation_amd.dropin as d  # noqa: E402

d.install()
from unet2d import UNetConfig, UNetModel  # noqa: E402

bs = int(sys.argv[1]) if len(sys.argv) > 1 else 4
torch.manual_seed(0)
m = UNetModel(UNetConfig(in_channels=3, out_channels=1, unet_type="UNet_3Plus")).cuda().train()
opt = torch.optim.AdamW(m.parameters(), lr=1e-4)
x = torch.randn(bs, 3, 512, 512, device="cuda")
t = (torch.rand(bs, 1, 512, 512, device="cuda") > 0.5).float()
def step():
    opt.zero_grad(set_to_none=True)
    out = m(images=x, labels=t)
    out.loss.backward()
    opt.step()
    return out.loss
for _ in range(2):
    step()
torch.cuda.synchronize()
t0 = time.perf_counter()
n = 5
for _ in range(n):
    loss = step()
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / n
print(f"UNet_3Plus {os.environ.get('MISAMD_DTYPE', 'f32')} bs={bs} 512x512: {dt * 1e3:.1f} ms/step = {bs / dt:.2f} images/s, loss {loss.item():.4f}, "
      f"peak mem {torch.cuda.max_memory_allocated() / 2**30:.1f} GiB")
