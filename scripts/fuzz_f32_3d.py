"""Randomised A/B of round 5's fp32 3x3x3 kernels (conv3d_f32.hip, wgrad_f32.hip) against the lock-step kernels they replace, same process, dispatch switches flipped:
ragged grids (down to one voxel), operands and destinations that are channel slices of wider buffers, ReLU and masked epilogues, padded-channel forward (Cin 32).
The convolution must be BIT-identical (same summation order), the weight gradient agrees to fp32 rounding (different split-K partition).
Round 6: 32-column tiles (Cout = 32 mod 64: against the same weights zero-padded to 64 columns on the 64-column kernel, bit-identical), 32-input-channel weight-gradient blocks
(Cin = 32 mod 64, against the lock-step kernel) and the statistics epilogue (MisConvDesc.st_mode 1 with one or two x sources, mode 2: the output must not move, the sums within
1e-6 of the sum of magnitudes of an fp64 evaluation).
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
    Cin = rng.choice([32, 64, 64, 96, 128, 192, 256, 384])
    Cout = rng.choice([32, 64, 64, 96, 128, 192, 256])
    narrow = Cout % 64 != 0
    Cp = (Cout + 63) // 64 * 64
    grid = (N, D, H, W)
    xoff, yoff = rng.choice([0, 64]), rng.choice([0, 64])
    xbuf = torch.randn(*grid, Cin + xoff + rng.choice([0, 64]), device=dev, generator=g)
    x = ops.View(xbuf, xoff, Cin)
    w = torch.randn(Cout, Cin, 3, 3, 3, device=dev, generator=g) * (27 * Cin) ** -0.5
    wf = torch.empty(27, Cout, Cin, device=dev)
    ops.pack_conv_weight(w, wf, None)
    if narrow:          # the reference launch: the same weights zero-padded to the next multiple of 64 columns
        wp = torch.zeros(Cp, Cin, 3, 3, 3, device=dev)
        wp[:Cout] = w
        wfp = torch.empty(27, Cp, Cin, device=dev)
        ops.pack_conv_weight(wp, wfp, None)
    mbuf = torch.randn(*grid, Cp + 64, device=dev, generator=g)
    form = rng.choice(["relu", "plain", "mask"])
    kw_new = dict(relu=True) if form == "relu" else (dict(mask=ops.View(mbuf, 64, Cout)) if form == "mask" else {})
    kw_old = dict(mask=ops.View(mbuf, 64, Cp)) if (narrow and form == "mask") else kw_new
    outs = []
    for old in (0, 1):
        co = Cp if (old and narrow) else Cout
        kw = kw_old if old else kw_new
        ybuf = torch.full((*grid, co + yoff), float("nan"), device=dev)
        with ops.dispatch_switches(MIS_CONV3D_F32_NOPP=(old and not narrow), MIS_CONV3D_F32_WIDE=case & 1):          # (odd cases: 128-column tiles even on grids that do not fill the chip)
            ops.conv_igemm(x, wfp if (old and narrow) else wf, ops.View(ybuf, yoff, co), ksize=3, Cin=Cin, Cout=co, grid=grid, **kw)
            tag = ops.conv_last_dispatch()
        if not old:
            tags[tag] = tags.get(tag, 0) + 1
            assert tag.startswith("k3.3d.f32pp"), tag
        if yoff and not torch.isnan(ybuf[..., :yoff]).all():
            print("WROTE OUTSIDE THE SLICE", grid, Cin, Cout, form, tag)
            bad += 1
        outs.append(ybuf[..., yoff:yoff + Cout])
    if narrow:
        assert "k3.3d.f32pp32" in tags, tags
    if not torch.equal(outs[0], outs[1]):
        bad += 1
        print("CONV MISMATCH", grid, Cin, Cout, form, (outs[0] - outs[1]).abs().max().item(), flush=True)
    # ---- statistics epilogue on the new kernels: the output must not move, the sums are those of the stored output ----
    two = (not narrow) and Cout >= 128 and rng.random() < 0.5 and D % 2 == 0 and H % 2 == 0 and W % 2 == 0
    S1, S2 = torch.full((N, Cout), float("nan"), device=dev), torch.full((N, Cout), float("nan"), device=dev)
    if two:
        C0 = 64 * rng.randrange(1, Cout // 64)
        g0 = torch.randn(*grid, C0, device=dev, generator=g)
        g1 = torch.randn(N, D // 2, H // 2, W // 2, Cout - C0, device=dev, generator=g)
        xs = torch.cat((g0, g1.repeat_interleave(2, 1).repeat_interleave(2, 2).repeat_interleave(2, 3)), -1)
        st = dict(mode=1, x0=g0, x1=g1, up=True, S1=S1, S2=S2)
    else:
        xs = torch.randn(*grid, Cout, device=dev, generator=g)
        st = dict(mode=rng.choice([1, 2]), x0=xs, S1=S1, S2=S2)
    ys = torch.full((*grid, Cout), float("nan"), device=dev)
    with ops.dispatch_switches(MIS_CONV3D_F32_WIDE=case & 1):
        ops.conv_igemm(x, wf, ys, ksize=3, Cin=Cin, Cout=Cout, grid=grid, stats=st, **kw_new)
    if not torch.equal(ys, outs[0]):
        bad += 1
        print("STATS EPILOGUE MOVED THE OUTPUT", grid, Cin, Cout, form, flush=True)
    yd = ys.double().view(N, -1, Cout)
    t2 = yd * (xs.double().view(N, -1, Cout) if st["mode"] == 1 else yd)
    for got, terms, what in ((S1, yd, "S1"), (S2, t2, "S2")):
        err = (got.double() - terms.sum(1)).abs()
        if not bool((err <= 1e-6 * terms.abs().sum(1) + 1e-30).all()):
            bad += 1
            print("STATS MISMATCH", what, grid, Cin, Cout, "mode", st["mode"], "two" if two else "one", (err / (terms.abs().sum(1) + 1e-30)).max().item(), flush=True)
    tags["stats.mode%d%s" % (st["mode"], ".two" if two else "")] = tags.get("stats.mode%d%s" % (st["mode"], ".two" if two else ""), 0) + 1
    if Cin % 32 == 0 and not narrow:
        dy = torch.randn(*grid, Cout, device=dev, generator=g)
        dws = []
        for old in (0, 1):
            dw = torch.full((Cout, Cin, 3, 3, 3), float("nan"), device=dev)
            with ops.dispatch_switches(MIS_WGRAD_F32_NOPP=old):
                ops.wgrad(x, dy, dw, ksize=3, Cin=Cin, Cout=Cout, grid=grid)
                tag, ns = ops.wgrad_last_dispatch()
            if not old:
                tags[tag] = tags.get(tag, 0) + 1
                assert tag == ("k3.3d.f32s" if Cin % 64 == 0 else "k3.3d.f32s32"), tag
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
