"""Per-layer timing of the fp32 3x3x3 kernels at cfg4's shapes (UNet3D(1,3), 2 x 128^3): forward conv, dgrad and weight gradient of every SingleConv, on the round-5
all-DMA kernels (conv3d_f32.hip / wgrad_f32.hip) and, with --old, on the lock-step kernels they replace (same process, dispatch override).
    python scripts/bench_f32_3d_layers.py [--size 128] [--old] [--iters 3]
Prints ms and TFLOP/s per launch and the three sums per train step; the f32 MFMA peak is 157.3 TFLOP/s."""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mdeical_image_segmentation_amd import ops  # noqa: E402

LAYERS = [  # (name, level, Cin operand, Cout, real Cin of the forward)
    ("enc0.c2", 0, 64, 64, 32), ("enc1.c1", 1, 64, 64, 64), ("enc1.c2", 1, 64, 128, 64), ("enc2.c1", 2, 128, 128, 128), ("enc2.c2", 2, 128, 256, 128),
    ("enc3.c1", 3, 256, 256, 256), ("enc3.c2", 3, 256, 512, 256), ("dec0.c1", 2, 768, 256, 768), ("dec0.c2", 2, 256, 256, 256),
    ("dec1.c1", 1, 384, 128, 384), ("dec1.c2", 1, 128, 128, 128), ("dec2.c1", 0, 192, 64, 192), ("dec2.c2", 0, 64, 64, 64),
]


def timed(fn, iters):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--size", type=int, default=128)
    ap.add_argument("--batch", type=int, default=2)
    ap.add_argument("--iters", type=int, default=3)
    ap.add_argument("--old", action="store_true")
    ap.add_argument("--only", default="")
    a = ap.parse_args()
    ops.load()
    dev = "cuda"
    sw = dict(MIS_CONV3D_F32_NOPP=1, MIS_WGRAD_F32_NOPP=1) if a.old else {}
    tot = {"fwd": 0.0, "dgrad": 0.0, "wgrad": 0.0}
    flops = {"fwd": 0.0, "dgrad": 0.0, "wgrad": 0.0}
    g = torch.Generator(device=dev).manual_seed(1)
    with ops.dispatch_switches(**sw):
        for name, lvl, cin, cout, cin_real in LAYERS:
            if a.only and a.only not in name:
                continue
            S = a.size >> lvl
            grid = (a.batch, S, S, S)
            x = torch.randn(*grid, cin, device=dev, generator=g)
            dy = torch.randn(*grid, cout, device=dev, generator=g)
            y = torch.empty(*grid, cout, device=dev)
            dx = torch.empty(*grid, cin, device=dev)
            w = torch.randn(cout, cin, 3, 3, 3, device=dev, generator=g) * (27 * cin) ** -0.5
            wf = torch.empty(27, cout, cin, device=dev)
            wd = torch.empty(27, cin, cout, device=dev)
            ops.pack_conv_weight(w, wf, wd)
            wfr = wf
            if cin_real != cin:
                wfr = torch.empty(27, cout, cin_real, device=dev)
                ops.pack_conv_weight(w[:, :cin_real].contiguous(), wfr, None)
            dw = torch.empty_like(w)
            npx = a.batch * S ** 3
            f_fwd = 2.0 * npx * 27 * cin_real * cout
            f_full = 2.0 * npx * 27 * cin * cout
            t_f = timed(lambda: ops.conv_igemm(ops.View(x, 0, cin_real) if cin_real != cin else x, wfr, y, ksize=3, Cin=cin_real, Cout=cout, grid=grid, relu=True), a.iters)
            c_f = ops.conv_last_dispatch()
            t_d = timed(lambda: ops.conv_igemm(dy, wd, dx, ksize=3, Cin=cout, Cout=cin, grid=grid), a.iters)
            c_d = ops.conv_last_dispatch()
            t_w = timed(lambda: ops.wgrad(x, dy, dw, ksize=3, Cin=cin, Cout=cout, grid=grid), a.iters)
            c_w, ns = ops.wgrad_last_dispatch()
            print(f"{name:8s} {cin:4d}->{cout:4d} @{S:3d}^3  fwd {t_f:7.3f} ms {f_fwd / t_f * 1e-9:6.1f} TF [{c_f}]  dgrad {t_d:7.3f} ms {f_full / t_d * 1e-9:6.1f} TF [{c_d}]  "
                  f"wgrad {t_w:7.3f} ms {f_full / t_w * 1e-9:6.1f} TF [{c_w} x{ns}]", flush=True)
            for k, t, f in (("fwd", t_f, f_fwd), ("dgrad", t_d, f_full), ("wgrad", t_w, f_full)):
                tot[k] += t
                flops[k] += f
            del x, dy, y, dx, w, wf, wd, dw
            torch.cuda.empty_cache()
    for k in tot:
        if tot[k] > 0:
            print(f"sum {k:5s}: {tot[k]:8.2f} ms  {flops[k] / tot[k] * 1e-9:6.1f} TFLOP/s = {flops[k] / tot[k] * 1e-9 / 157.3:.3f} of the f32 MFMA peak")
    t = sum(tot.values())
    if t > 0:
        print(f"sum all  : {t:8.2f} ms  {sum(flops.values()) / t * 1e-9:6.1f} TFLOP/s = {sum(flops.values()) / t * 1e-9 / 157.3:.3f}")


if __name__ == "__main__":
    main()
