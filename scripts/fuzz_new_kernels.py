"""Randomised A/B of the kernels added at the end of round 3 against the kernels they replace (same operands, dispatch switches flipped in one process):
streaming weight gradients vs the tile-staged row kernels, the ping-pong 1x1 weight gradient vs wgrad_kernel, the ping-pong 1x1 GEMM vs the generic configurations,
the deep-prefetch 64-column conv vs conv_ppc_kernel<8, 2>.  python scripts/fuzz_new_kernels.py [cases] [seed]"""
import os
import random
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mdeical_image_segmentation_amd import ops  # noqa: E402

ncases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
dev = "cuda"
BF = torch.bfloat16
worst = {}


def rel(a, b):
    return ((a.float() - b.float()).norm() / (b.float().norm() + 1e-30)).item()


count = {}


def note(kind, r, desc):
    count[kind] = count.get(kind, 0) + 1
    if kind not in worst or r > worst[kind][0]:
        worst[kind] = (r, desc)


for case in range(ncases):
    g = torch.Generator(device=dev).manual_seed(1000 + case)
    N = rng.choice([1, 2, 3, 5])
    H = 8 * rng.randint(1, 12)
    W = 32 * rng.randint(1, 5)
    Cin = rng.choice([64, 128, 192, 256])
    Cout = rng.choice([64, 128, 192, 256])
    x = torch.randn(N, H, W, Cin, device=dev, generator=g).to(BF)
    dy = torch.randn(N, H, W, Cout, device=dev, generator=g).to(BF)
    # --- 3x3 weight gradient: streaming vs row kernels
    dw0, dw1 = torch.empty(Cout, Cin, 3, 3, device=dev), torch.empty(Cout, Cin, 3, 3, device=dev)
    db0, db1 = torch.empty(Cout, device=dev), torch.empty(Cout, device=dev)
    ops.wgrad(x, dy, dw1, ksize=3, Cin=Cin, Cout=Cout, dbias=db1)
    t1 = ops.wgrad_last_dispatch()[0]
    ops.dispatch_override("MIS_WGRAD_PP_NOSTREAM", 1)
    ops.wgrad(x, dy, dw0, ksize=3, Cin=Cin, Cout=Cout, dbias=db0)
    t0 = ops.wgrad_last_dispatch()[0]
    ops.dispatch_override("MIS_WGRAD_PP_NOSTREAM", -1)
    note("wgrad3 " + t1, max(rel(dw1, dw0), rel(db1, db0)), f"{N}x{H}x{W} {Cin}->{Cout} vs {t0}")
    # --- 3x3 conv, 64-column blocks: deep prefetch vs conv_ppc_kernel<8, 2>
    if Cout % 128 != 0:
        w = (torch.randn(9, Cout, Cin, device=dev, generator=g) * (9 * Cin) ** -0.5).to(BF)
        b = torch.randn(Cout, device=dev, generator=g)
        y0, y1 = torch.empty(N, H, W, Cout, device=dev, dtype=BF), torch.empty(N, H, W, Cout, device=dev, dtype=BF)
        ops.dispatch_override("MIS_CONV_PPC64", 1)
        ops.conv_igemm(x, w, y1, ksize=3, Cin=Cin, Cout=Cout, bias=b, relu=True)
        c1 = ops.conv_last_dispatch()
        ops.dispatch_override("MIS_CONV_NOPPD", 1)
        ops.conv_igemm(x, w, y0, ksize=3, Cin=Cin, Cout=Cout, bias=b, relu=True)
        c0 = ops.conv_last_dispatch()
        ops.dispatch_override("MIS_CONV_NOPPD", -1)
        ops.dispatch_override("MIS_CONV_PPC64", -1)
        note("conv3 " + c1, 0.0 if torch.equal(y0, y1) else max(rel(y1, y0), 1e-9), f"{N}x{H}x{W} {Cin}->{Cout} vs {c0}")
    # --- 1x1: GEMM forward (pixel-shuffled) and weight gradient
    C1 = rng.choice([128, 256])
    Cq = rng.choice([64, 128])
    if (N * H * W) % 64 == 0 and H % 32 == 0:
        x1 = torch.randn(N, H, W, C1, device=dev, generator=g).to(BF)
        wf = (torch.randn(1, 4 * Cq, C1, device=dev, generator=g) * C1 ** -0.5).to(BF)
        bq = torch.randn(Cq, device=dev, generator=g)
        cat0 = torch.zeros(N, 2 * H, 2 * W, 2 * Cq, device=dev, dtype=BF)
        cat1 = torch.zeros_like(cat0)
        ops.conv_igemm(x1, wf, ops.View(cat1, 0, Cq), ksize=1, Cin=C1, Cout=4 * Cq, bias=bq, y0_mode=ops.OUT_SHUFFLE2)
        k1 = ops.conv_last_dispatch()
        ops.dispatch_override("MIS_GEMM1_NOPP", 1)
        ops.conv_igemm(x1, wf, ops.View(cat0, 0, Cq), ksize=1, Cin=C1, Cout=4 * Cq, bias=bq, y0_mode=ops.OUT_SHUFFLE2)
        k0 = ops.conv_last_dispatch()
        ops.dispatch_override("MIS_GEMM1_NOPP", -1)
        note("gemm1 " + k1, 0.0 if torch.equal(cat0, cat1) else max(rel(cat1, cat0), 1e-9), f"{N}x{H}x{W} {C1}->{4 * Cq} vs {k0}")
        dys = torch.randn(N, H, W, 4 * Cq, device=dev, generator=g).to(BF)
        dwa, dwb = torch.empty(C1, Cq, 2, 2, device=dev), torch.empty(C1, Cq, 2, 2, device=dev)
        dba, dbb = torch.empty(Cq, device=dev), torch.empty(Cq, device=dev)
        ops.wgrad(x1, dys, dwb, ksize=1, Cin=C1, Cout=4 * Cq, dw_layout=1, dbias=dbb)
        g1 = ops.wgrad_last_dispatch()[0]
        ops.dispatch_override("MIS_WGRAD_K1_NOPP", 1)
        ops.wgrad(x1, dys, dwa, ksize=1, Cin=C1, Cout=4 * Cq, dw_layout=1, dbias=dba)
        g0 = ops.wgrad_last_dispatch()[0]
        ops.dispatch_override("MIS_WGRAD_K1_NOPP", -1)
        note("wgrad1 " + g1, max(rel(dwb, dwa), rel(dbb, dba)), f"{N}x{H}x{W} {C1}->{4 * Cq} vs {g0}")
# --- 3-D weight gradients: streaming vs tile-staged row kernels (planes above / below the volume, kd in the block identity)
for case in range(max(ncases // 4, 4)):
    g = torch.Generator(device=dev).manual_seed(5000 + case)
    N, D = rng.choice([1, 2]), rng.randint(1, 6)
    H, W = 8 * rng.randint(1, 5), 32 * rng.randint(1, 3)
    Cin, Cout = rng.choice([64, 128, 192]), rng.choice([64, 128])
    x = torch.randn(N, D, H, W, Cin, device=dev, generator=g).to(BF)
    dy = torch.randn(N, D, H, W, Cout, device=dev, generator=g).to(BF)
    dw0, dw1 = torch.empty(Cout, Cin, 3, 3, 3, device=dev), torch.empty(Cout, Cin, 3, 3, 3, device=dev)
    ops.wgrad(x, dy, dw1, ksize=3, Cin=Cin, Cout=Cout, grid=(N, D, H, W))
    t1 = ops.wgrad_last_dispatch()[0]
    ops.dispatch_override("MIS_WGRAD_PP_NOSTREAM", 1)
    ops.wgrad(x, dy, dw0, ksize=3, Cin=Cin, Cout=Cout, grid=(N, D, H, W))
    t0 = ops.wgrad_last_dispatch()[0]
    ops.dispatch_override("MIS_WGRAD_PP_NOSTREAM", -1)
    note("wgrad3 " + t1, rel(dw1, dw0), f"{N}x{D}x{H}x{W} {Cin}->{Cout} vs {t0}")
torch.cuda.synchronize()
bad = 0
for k, (r, d) in sorted(worst.items()):
    print(f"{k:24s} {count[k]:3d} cases, worst rel. L2 difference {r:.3e}   ({d})")
    bad += r > 2e-5 if k.startswith("wgrad") else r > 2e-2
print("FUZZ", "FAILED" if bad else "ok", f"({ncases} cases)")
sys.exit(1 if bad else 0)
