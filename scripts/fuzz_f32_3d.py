"""Randomised A/B of round 5's fp32 3x3x3 kernels (conv3d_f32.hip, wgrad_f32.hip) against the lock-step kernels they replace, same process, dispatch switches flipped:
ragged grids (down to one voxel), operands and destinations that are channel slices of wider buffers, ReLU and masked epilogues, padded-channel forward (Cin 32).
The convolution must be BIT-identical (same summation order), the weight gradient agrees to fp32 rounding (different split-K partition).
    python scripts/fuzz_f32_3d.py [cases] [seed]"""
import os
import random
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mdeical_image_segmentation_amd import ops  # noqa: E402

ncases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
rng = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
dev = "cuda"
ops.load()
bad = 0
worst_w = (0.0, "")
tags = {}
for case in range(ncases):
    g = torch.Generator(device=dev).manual_seed(5000 + case)
    N = rng.choice([1, 1, 2, 3])
    D = rng.choice([1, 2, 3, 5, 8, 9])
    H = rng.choice([1, 3, 7, 8, 9, 16, 17, 24, 33, 44])
    W = rng.choice([1, 5, 15, 16, 17, 31, 32, 33, 48, 52])
    Cin = rng.choice([32, 64, 64, 128, 192, 256, 384])
    Cout = rng.choice([64, 64, 128, 192, 256])
    grid = (N, D, H, W)
    xoff, yoff = rng.choice([0, 64]), rng.choice([0, 64])
    xbuf = torch.randn(*grid, Cin + xoff + rng.choice([0, 64]), device=dev, generator=g)
    x = ops.View(xbuf, xoff, Cin)
    w = torch.randn(Cout, Cin, 3, 3, 3, device=dev, generator=g) * (27 * Cin) ** -0.5
    wf = torch.empty(27, Cout, Cin, device=dev)
    ops.pack_conv_weight(w, wf, None)
    mbuf = torch.randn(*grid, Cout + 64, device=dev, generator=g)
    form = rng.choice(["relu", "plain", "mask"])
    kw = dict(relu=True) if form == "relu" else (dict(mask=ops.View(mbuf, 64, Cout)) if form == "mask" else {})
    outs = []
    for old in (0, 1):
        ybuf = torch.full((*grid, Cout + yoff), float("nan"), device=dev)
        with ops.dispatch_switches(MIS_CONV3D_F32_NOPP=old, MIS_CONV3D_F32_WIDE=case & 1):          # (odd cases: 128-column tiles even on grids that do not fill the chip)
            ops.conv_igemm(x, wf, ops.View(ybuf, yoff, Cout), ksize=3, Cin=Cin, Cout=Cout, grid=grid, **kw)
            tag = ops.conv_last_dispatch()
        if not old:
            tags[tag] = tags.get(tag, 0) + 1
            assert tag.startswith("k3.3d.f32pp"), tag
        if yoff and not torch.isnan(ybuf[..., :yoff]).all():
            print("WROTE OUTSIDE THE SLICE", grid, Cin, Cout, form, tag)
            bad += 1
        outs.append(ybuf[..., yoff:])
    if not torch.equal(outs[0], outs[1]):
        bad += 1
        print("CONV MISMATCH", grid, Cin, Cout, form, (outs[0] - outs[1]).abs().max().item(), flush=True)
    if Cin % 64 == 0:
        dy = torch.randn(*grid, Cout, device=dev, generator=g)
        dws = []
        for old in (0, 1):
            dw = torch.full((Cout, Cin, 3, 3, 3), float("nan"), device=dev)
            with ops.dispatch_switches(MIS_WGRAD_F32_NOPP=old):
                ops.wgrad(x, dy, dw, ksize=3, Cin=Cin, Cout=Cout, grid=grid)
                tag, ns = ops.wgrad_last_dispatch()
            if not old:
                tags[tag] = tags.get(tag, 0) + 1
                assert tag == "k3.3d.f32s", tag
            dws.append(dw)
        r = ((dws[0] - dws[1]).norm() / (dws[1].norm() + 1e-30)).item()
        if not (r < 2e-6) or not torch.isfinite(dws[0]).all():
            bad += 1
            print("WGRAD MISMATCH", grid, Cin, Cout, r, flush=True)
        if r > worst_w[0]:
            worst_w = (r, f"{grid} {Cin}->{Cout} nsplit {ns}")
    if case % 10 == 9:
        print(f"{case + 1} cases, {bad} bad", flush=True)
print("kernels hit:", tags)
print("worst wgrad rel-L2 vs the lock-step kernel:", worst_w)
print("FUZZ", "FAILED" if bad else "OK", f"({ncases} cases)")
sys.exit(1 if bad else 0)
