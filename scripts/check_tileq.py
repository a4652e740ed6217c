"""Dynamic tile queue (MIS_TILEQ_OFF=0, default) against the static stride (MIS_TILEQ_OFF=1) on the persistent conv kernels: bit-identical outputs (the tiles are the
same, only who runs them changes), the counters back at zero after every launch, and ms per launch both ways.  python scripts/check_tileq.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mdeical_image_segmentation_amd import ops  # noqa: E402

ops.load()
dev = "cuda"
BF = torch.bfloat16
bad = 0


def timed(fn, iters=8):
    best = 1e9
    for _ in range(3):
        fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            fn()
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / iters)
    return best


g = torch.Generator(device=dev).manual_seed(3)
for N, H, W, Cin, Cout in ((32, 64, 64, 512, 512), (32, 256, 256, 128, 128), (32, 512, 512, 64, 128), (32, 32, 32, 1024, 1024), (3, 150, 170, 64, 256), (5, 150, 170, 128, 128),
                           (1, 32, 16, 64, 128), (32, 512, 512, 64, 64), (32, 512, 512, 128, 64), (7, 100, 90, 192, 64)):
    x = torch.randn(N, H, W, Cin, device=dev, generator=g).to(BF)
    w = (torch.randn(9, Cout, Cin, device=dev, generator=g) * (9 * Cin) ** -0.5).to(BF)
    b = torch.randn(Cout, device=dev, generator=g)
    m = torch.randn(N, H, W, Cout, device=dev, generator=g).to(BF)
    for form, kw in (("fwd", dict(bias=b, relu=True)), ("mask", dict(mask=m))):
        outs, ts = [], []
        for off in (1, 0):
            y = torch.full((N, H, W, Cout), float("nan"), device=dev, dtype=BF)
            with ops.dispatch_switches(MIS_TILEQ_OFF=off):
                fn = lambda: ops.conv_igemm(x, w, y, ksize=3, Cin=Cin, Cout=Cout, **kw)      # noqa: E731
                ts.append(timed(fn))
                tag = ops.conv_last_dispatch()
            outs.append(y)
        same = torch.equal(outs[0], outs[1])
        bad += 0 if same else 1
        print(f"{N:3d}x{H:3d}x{W:3d} {Cin:4d}->{Cout:<4d} {form:4s} [{tag}] static {ts[0]:6.3f} ms  queue {ts[1]:6.3f} ms  ({(ts[1] / ts[0] - 1) * 100:+5.1f} %)  bit-identical: {same}", flush=True)
# 3-D bf16 (conv3d_ppc_kernel): plain and masked forms
for grid, Cin, Cout in (((2, 16, 80, 80), 128, 128), ((1, 24, 160, 160), 192, 64), ((2, 10, 40, 40), 256, 256), ((1, 7, 33, 21), 64, 192), ((1, 20, 160, 160), 64, 64)):
    x = torch.randn(*grid, Cin, device=dev, generator=g).to(BF)
    w = (torch.randn(27, Cout, Cin, device=dev, generator=g) * (27 * Cin) ** -0.5).to(BF)
    m = torch.randn(*grid, Cout, device=dev, generator=g).to(BF)
    for form, kw in (("relu", dict(relu=True)), ("mask", dict(mask=m))):
        outs, ts = [], []
        for off in (1, 0):
            y = torch.full((*grid, Cout), float("nan"), device=dev, dtype=BF)
            with ops.dispatch_switches(MIS_TILEQ_OFF=off):
                fn = lambda: ops.conv_igemm(x, w, y, ksize=3, Cin=Cin, Cout=Cout, grid=grid, **kw)      # noqa: E731
                ts.append(timed(fn, 4))
                tag = ops.conv_last_dispatch()
            outs.append(y)
        same = torch.equal(outs[0], outs[1])
        bad += 0 if same else 1
        print(f"{'x'.join(map(str, grid)):>14s} {Cin:4d}->{Cout:<4d} {form:4s} [{tag}] static {ts[0]:6.3f} ms  queue {ts[1]:6.3f} ms  ({(ts[1] / ts[0] - 1) * 100:+5.1f} %)  bit-identical: {same}", flush=True)
print("TILEQ", "FAILED" if bad else "OK")
sys.exit(1 if bad else 0)
