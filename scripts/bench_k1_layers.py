"""The 1x1 GEMMs of the 2-D benchmark net (ConvTranspose2d(k2, s2) forward with the pixel-shuffle epilogue, its dgrad with the ReLU mask, its weight gradient):
bs 32, bf16, per-call time and the HBM rate of the algorithmic bytes.  Dispatcher switches for these kernels are read once per process: run once per arm.

    python scripts/bench_k1_layers.py [NAME=VALUE ...]"""
import os
import sys

import torch

for a in sys.argv[1:]:
    k, v = a.split("=")
    os.environ[k] = v
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mdeical_image_segmentation_amd import ops  # noqa: E402
from mdeical_image_segmentation_amd._lib import OUT_SHUFFLE2  # noqa: E402
from mdeical_image_segmentation_amd.ops import View  # noqa: E402

N, REP, ROUNDS = 32, 5, 3
dev = "cuda"
tot = 0.0
for h, c in ((32, 512), (64, 256), (128, 128), (256, 64)):
    g = torch.Generator(device=dev).manual_seed(1)
    x = torch.randn(N, h, h, 2 * c, device=dev, generator=g).relu().to(torch.bfloat16)
    wf = (torch.randn(1, 4 * c, 2 * c, device=dev, generator=g) * (2 * c) ** -0.5).to(torch.bfloat16)
    wd = (torch.randn(1, 2 * c, 4 * c, device=dev, generator=g) * (4 * c) ** -0.5).to(torch.bfloat16)
    b = torch.randn(c, device=dev, generator=g)
    cat = torch.empty(N, 2 * h, 2 * h, 2 * c, device=dev, dtype=torch.bfloat16)
    dys = torch.randn(N, h, h, 4 * c, device=dev, generator=g).to(torch.bfloat16)
    gin = torch.empty(N, h, h, 2 * c, device=dev, dtype=torch.bfloat16)
    dw = torch.empty(2 * c, c, 2, 2, device=dev)
    db = torch.empty(c, device=dev)
    calls = {
        "fwd": lambda: ops.conv_igemm(x, wf, View(cat, 0, c), ksize=1, Cin=2 * c, Cout=4 * c, bias=b, relu=False, y0_mode=OUT_SHUFFLE2),
        "dgrad": lambda: ops.conv_igemm(dys, wd, gin, ksize=1, Cin=4 * c, Cout=2 * c, mask=x),
        "wgrad": lambda: ops.wgrad(x, dys, dw, ksize=1, Cin=2 * c, Cout=4 * c, dw_layout=1, dbias=db),
    }
    px = N * h * h
    nbytes = {"fwd": px * (2 * c + 4 * c) * 2, "dgrad": px * (4 * c + 2 * c + 2 * c) * 2, "wgrad": px * (2 * c + 4 * c) * 2}
    for name, fn in calls.items():
        fn()
        tag = ops.conv_last_dispatch() if name != "wgrad" else ops.wgrad_last_dispatch()[0]
        best = 1e9
        for _ in range(ROUNDS):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(REP):
                fn()
            e1.record()
            torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) / REP)
        tot += best
        print(f"{h:4d}^2 {2 * c:5d}->{4 * c:<5d} {name:6s} {best:7.3f} ms  {nbytes[name] / best / 1e9:6.2f} TB/s  {2.0 * px * 2 * c * 4 * c / best / 1e9:6.0f} TF/s  {tag}", flush=True)
print(f"per-step total: {tot:.3f} ms   ({' '.join(sys.argv[1:]) or 'default'})")
