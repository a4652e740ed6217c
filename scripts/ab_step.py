"""In-process A/B of the whole 2-D train step (bs 32, 512^2, bf16) under dispatcher switches (set through mis_dispatch_override) (cdna_hip_programming.md §5.4 rule 24: boxes differ by
several per cent, so arms are interleaved in ONE process):   python scripts/ab_step.py MIS_CONV_NOPPC=1 [NAME=VALUE ...]     (arm 0 = no switch)"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mdeical_image_segmentation_amd import ops  # noqa: E402
from mdeical_image_segmentation_amd.engine2d import UNet2DEngine  # noqa: E402

arms = [("default", {})] + [(a, dict([a.split("=")])) for a in sys.argv[1:]]
eng = UNet2DEngine(1, 2, dtype=torch.bfloat16, device="cuda", seed=0, lr=1e-5)
g = torch.Generator().manual_seed(1000)
x = torch.randn(32, 1, 512, 512, generator=g).cuda()
y = torch.randint(0, 2, (32, 512, 512), generator=g).cuda()
for _ in range(3):
    eng.train_step(x, y)
best = {n: 1e9 for n, _ in arms}
for r in range(4):
    for name, env in arms:
        for k, v in env.items():
            ops.dispatch_override(k, int(v))
        eng.train_step(x, y)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(8):
            eng.train_step(x, y)
        torch.cuda.synchronize()
        best[name] = min(best[name], (time.perf_counter() - t0) / 8 * 1e3)
        for k in env:
            ops.dispatch_override(k, -1)
for name, _ in arms:
    print(f"{name:28s} {best[name]:7.3f} ms/step  {32e3 / best[name]:7.1f} img/s")
