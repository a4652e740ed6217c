"""Tile-height probe for the bf16 3x3x3 ping-pong kernels (conv3d_pp.hip): the same layer with 20- / 32- / 40-row tiles (MIS_CONV3D_PF = 5 / 8 / 10) on grids whose
H axis all three divide (H = 160), i.e. the only thing that changes is the tile: LDS-DMA instructions and fragment reads per MFMA.
    python scripts/bench_3d_pf.py [D]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mdeical_image_segmentation_amd import ops  # noqa: E402

D = int(sys.argv[1]) if len(sys.argv) > 1 else 16
ops.load()
for Cin, Cout, pfs in ((128, 128, (5, 8, 10)), (384, 128, (5, 8, 10)), (256, 256, (5, 8, 10)), (64, 64, (5, 8, 10)), (192, 64, (5, 8, 10))):
    g = torch.Generator(device="cuda").manual_seed(1)
    grid = (1, D, 160, 160)
    x = torch.randn(*grid, Cin, device="cuda", generator=g).to(torch.bfloat16)
    w = (torch.randn(27, Cout, Cin, device="cuda", generator=g) * (27 * Cin) ** -0.5).to(torch.bfloat16)
    y = torch.empty(*grid, Cout, device="cuda", dtype=torch.bfloat16)
    ref = None
    for pf in pfs:
        with ops.dispatch_switches(MIS_CONV3D_PF=pf):
            best = 1e9
            for r in range(3):
                ops.conv_igemm(x, w, y, ksize=3, Cin=Cin, Cout=Cout, grid=grid, relu=True)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(5):
                    ops.conv_igemm(x, w, y, ksize=3, Cin=Cin, Cout=Cout, grid=grid, relu=True)
                e1.record()
                torch.cuda.synchronize()
                best = min(best, e0.elapsed_time(e1) / 5)
            same = "" if ref is None else f"  bit-identical to PF {pfs[0]}: {bool(torch.equal(ref, y))}"
            if ref is None:
                ref = y.clone()
            print(f"{Cin:4d}->{Cout:<4d} {D}x160x160  PF {pf:2d}  {best:7.3f} ms {2.0 * D * 160 * 160 * 27 * Cin * Cout / best / 1e9:6.0f} TF  {ops.conv_last_dispatch()}{same}", flush=True)
