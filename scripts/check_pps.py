"""conv_pps_kernel (MIS_CONV_PPS=1, one wave per SIMD) against conv_ppc_kernel<8, 4>: bit-identical outputs (same per-accumulator summation order) on a set of shapes in the
forward, bf16-mask and ReLU-bits forms, then time per launch of both on the benchmark's layer classes: python scripts/check_pps.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mdeical_image_segmentation_amd import ops  # noqa: E402

BF = torch.bfloat16
dev = "cuda"
SW = os.environ.get("CHECK_SWITCH", "MIS_CONV_PPS")          # the kernel under test: MIS_CONV_PPS | MIS_CONV_PPC2


def run(x, w, b, form, pps, Cin, Cout, m=None, bits=None):
    y = torch.full((*x.shape[:3], Cout), float("nan"), dtype=BF, device=dev)
    kw = dict(bias=b, relu=True) if form == "fwd" else (dict(mask=m) if form == "mask" else dict(mask_bits=bits))
    with ops.dispatch_switches(**{SW: 1 if pps else 0}):
        ops.conv_igemm(x, w, y, ksize=3, Cin=Cin, Cout=Cout, **kw)
        tag = ops.conv_last_dispatch()
    return y, tag


def main():
    ops.load()
    g = torch.Generator(device=dev).manual_seed(1)
    bad = 0
    for (N, H, W, Cin, Cout) in [(2, 64, 64, 128, 128), (1, 32, 16, 64, 128), (3, 150, 170, 64, 256), (2, 20, 36, 256, 256), (1, 9, 17, 1024, 1024), (5, 150, 170, 128, 128),
                                 (2, 100, 150, 128, 256), (1, 70, 90, 192, 128)]:
        x = torch.randn(N, H, W, Cin, device=dev, generator=g).to(BF)
        w = (torch.randn(9, Cout, Cin, device=dev, generator=g) * (9 * Cin) ** -0.5).to(BF)
        b = torch.randn(Cout, device=dev, generator=g)
        m = torch.randn(N, H, W, Cout, device=dev, generator=g).to(BF)
        bits = torch.empty(ops.relu_bits_bytes(N, H, W, Cout), dtype=torch.uint8, device=dev)
        ops.relu_bits(m, bits) if hasattr(ops, "relu_bits") else None
        for form in ("fwd", "mask") + (("bits",) if hasattr(ops, "relu_bits") else ()):
            ya, ta = run(x, w, b, form, False, Cin, Cout, m, bits)
            yb, tb = run(x, w, b, form, True, Cin, Cout, m, bits)
            same = torch.equal(ya, yb)
            bad += not same
            print(f"{N}x{H}x{W} {Cin}->{Cout} {form:4s} {ta:16s} vs {tb:16s} {'bit-identical' if same else 'DIFF max ' + str((ya.float() - yb.float()).abs().max().item())}", flush=True)
    print("mismatches:", bad)
    for H, Cin, Cout in [(64, 512, 512), (32, 1024, 1024), (128, 256, 256), (256, 128, 128), (64, 1024, 512), (512, 64, 128), (256, 64, 128)]:
        x = torch.randn(32, H, H, Cin, device=dev, generator=g).to(BF)
        w = (torch.randn(9, Cout, Cin, device=dev, generator=g) * (9 * Cin) ** -0.5).to(BF)
        b = torch.randn(Cout, device=dev, generator=g)
        y = torch.empty(32, H, H, Cout, dtype=BF, device=dev)
        res = []
        for pps in (0, 1, 0, 1):
            with ops.dispatch_switches(**{SW: pps}):
                ops.conv_igemm(x, w, y, ksize=3, Cin=Cin, Cout=Cout, bias=b, relu=True)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(5):
                    ops.conv_igemm(x, w, y, ksize=3, Cin=Cin, Cout=Cout, bias=b, relu=True)
                e1.record()
                torch.cuda.synchronize()
                res.append(e0.elapsed_time(e1) / 5)
        fl = 2.0 * 32 * H * H * 9 * Cin * Cout
        print(f"{H:4d}^2 {Cin:5d}->{Cout:<5d} ppc {min(res[0], res[2]):7.3f} ms {fl / min(res[0], res[2]) / 1e9:6.0f} TF   pps {min(res[1], res[3]):7.3f} ms {fl / min(res[1], res[3]) / 1e9:6.0f} TF", flush=True)


if __name__ == "__main__":
    main()
