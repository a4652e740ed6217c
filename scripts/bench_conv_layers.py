"""A/B of mis_conv_igemm configurations on the 3x3 layer shapes of the 2-D benchmark net (bs 32, 512^2, bf16, random data), interleaved rounds in ONE
process (cdna_hip_programming.md §5.4 rule 24).  Arms are dispatcher switches set through mis_dispatch_override.

    python scripts/bench_conv_layers.py                       # column-segment kernel vs conv_pp_kernel (MIS_CONV_NOPPC=1) vs the round-1 kernels (MIS_CONV_NOPP=1)
    python scripts/bench_conv_layers.py MIS_CONV_PP_NO256=1   # extra arm(s): NAME=VALUE"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mdeical_image_segmentation_amd import ops  # noqa: E402

LAYERS = [  # (H, Cin, Cout, launches per step with this shape incl. dgrads)
    (512, 64, 128, 1), (256, 128, 128, 4), (256, 128, 256, 1), (256, 256, 128, 1), (128, 256, 256, 4), (128, 256, 512, 1), (128, 512, 256, 1),
    (64, 512, 512, 4), (64, 512, 1024, 1), (64, 1024, 512, 1), (32, 1024, 1024, 2), (32, 512, 1024, 1), (32, 1024, 512, 1),
    (256, 64, 128, 1), (128, 128, 256, 1), (128, 256, 128, 1), (64, 256, 512, 1), (64, 512, 256, 1),
    (512, 64, 64, 4), (512, 128, 64, 1), (256, 128, 64, 1),
]
arms = [("ppc", {}), ("pp", {"MIS_CONV_NOPPC": "1"}), ("nopp", {"MIS_CONV_NOPP": "1"})]
for a in sys.argv[1:]:
    k, v = a.split("=")
    arms.append((a, {k: v}))
N, ROUNDS, REP = 32, 3, 3
dev = "cuda"
tot = {name: 0.0 for name, _ in arms}
print(f"{'layer':28s} " + " ".join(f"{name:>22s}" for name, _ in arms))
for H, Cin, Cout, mult in LAYERS:
    g = torch.Generator(device=dev).manual_seed(1)
    x = torch.randn(N, H, H, Cin, device=dev, generator=g).to(torch.bfloat16)
    w = (torch.randn(9, Cout, Cin, device=dev, generator=g) * (9 * Cin) ** -0.5).to(torch.bfloat16)
    b = torch.randn(Cout, device=dev, generator=g)
    y = torch.empty(N, H, H, Cout, device=dev, dtype=torch.bfloat16)
    flops = 2.0 * N * H * H * 9 * Cin * Cout
    best = {name: 1e9 for name, _ in arms}
    cfgs = {}
    for r in range(ROUNDS):
        for name, env in arms:
            for k, v in env.items():
                ops.dispatch_override(k, int(v))
            ops.conv_igemm(x, w, y, ksize=3, Cin=Cin, Cout=Cout, bias=b, relu=True)       # warm
            cfgs[name] = ops.conv_last_dispatch()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(REP):
                ops.conv_igemm(x, w, y, ksize=3, Cin=Cin, Cout=Cout, bias=b, relu=True)
            e1.record()
            torch.cuda.synchronize()
            best[name] = min(best[name], e0.elapsed_time(e1) / REP)
            for k in env:
                ops.dispatch_override(k, -1)
    for name, _ in arms:
        tot[name] += best[name] * mult
    print(f"{H:4d}^2 {Cin:5d}->{Cout:<5d} x{mult}      " + " ".join(f"{best[n]:7.3f}ms {flops / best[n] / 1e9:6.0f}TF {cfgs[n][6:]:>6s}" for n, _ in arms), flush=True)
print("per-step total (ms):      " + " ".join(f"{tot[n]:22.3f}" for n, _ in arms))
