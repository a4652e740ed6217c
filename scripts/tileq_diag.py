"""Small-grid diagnostic of the tile queue: static path, then the queue, counters printed after every launch.  python3 -u scripts/tileq_diag.py"""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mdeical_image_segmentation_amd import ops, _lib  # noqa: E402

lib = _lib.load()
dev = "cuda"
BF = torch.bfloat16


def counters():
    out = (ctypes.c_uint * 8)()
    rc = lib.mis_debug_tile_queue(ops.stream_ptr(), out)
    return rc, list(out)


g = torch.Generator(device=dev).manual_seed(3)
for N, H, W, Cin, Cout in ((16, 64, 64, 64, 512), (32, 64, 64, 512, 512)):
    x = torch.randn(N, H, W, Cin, device=dev, generator=g).to(BF)
    w = (torch.randn(9, Cout, Cin, device=dev, generator=g) * (9 * Cin) ** -0.5).to(BF)
    b = torch.randn(Cout, device=dev, generator=g)
    outs = []
    for off in (1, 0):
        y = torch.full((N, H, W, Cout), float("nan"), device=dev, dtype=BF)
        with ops.dispatch_switches(MIS_TILEQ_OFF=off):
            for it in range(3):
                print(f"{N}x{H}x{W} {Cin}->{Cout} off={off} launch {it} ...", flush=True)
                ops.conv_igemm(x, w, y, ksize=3, Cin=Cin, Cout=Cout, bias=b, relu=True)
                torch.cuda.synchronize()
                print("   done", ops.conv_last_dispatch(), "counters", counters() if not off else "-", "nan:", int(torch.isnan(y.float()).sum()), flush=True)
        outs.append(y)
    print("bit-identical:", torch.equal(outs[0], outs[1]), flush=True)
