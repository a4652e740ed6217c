"""A/B of mis_wgrad configurations on the 3x3 layer shapes of the 2-D benchmark net (bs 32, 512^2, bf16, random data), interleaved rounds in ONE process.
    python scripts/bench_wgrad_layers.py [NAME=VALUE ...]      # arms: default (ping-pong), MIS_WGRAD_NOPP=1, plus the given dispatcher switches"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mdeical_image_segmentation_amd import ops  # noqa: E402

LAYERS = [  # (H, Cin, Cout, launches per step)
    (512, 64, 64, 2), (512, 128, 64, 1), (256, 64, 128, 1), (256, 128, 128, 2), (256, 256, 128, 1), (128, 128, 256, 1), (128, 256, 256, 2),
    (128, 512, 256, 1), (64, 256, 512, 1), (64, 512, 512, 2), (64, 1024, 512, 1), (32, 512, 1024, 1), (32, 1024, 1024, 1),
]
arms = [("pp", {}), ("nopp", {"MIS_WGRAD_NOPP": "1"})]
for a in sys.argv[1:]:
    k, v = a.split("=")
    arms.append((a, {k: v}))
N, ROUNDS, REP = 32, 3, 3
dev = "cuda"
tot = {name: 0.0 for name, _ in arms}
print(f"{'layer':28s} " + " ".join(f"{name:>26s}" for name, _ in arms))
for H, Cin, Cout, mult in LAYERS:
    g = torch.Generator(device=dev).manual_seed(1)
    x = torch.randn(N, H, H, Cin, device=dev, generator=g).to(torch.bfloat16)
    dy = torch.randn(N, H, H, Cout, device=dev, generator=g).to(torch.bfloat16)
    dw = torch.empty(Cout, Cin, 3, 3, device=dev)
    db = torch.empty(Cout, device=dev)
    flops = 2.0 * N * H * H * 9 * Cin * Cout
    best = {name: 1e9 for name, _ in arms}
    cfgs = {}
    for r in range(ROUNDS):
        for name, env in arms:
            for k, v in env.items():
                ops.dispatch_override(k, int(v))
            ops.wgrad(x, dy, dw, ksize=3, Cin=Cin, Cout=Cout, dbias=db)
            cfgs[name] = "%s/%d" % ops.wgrad_last_dispatch()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(REP):
                ops.wgrad(x, dy, dw, ksize=3, Cin=Cin, Cout=Cout, dbias=db)
            e1.record()
            torch.cuda.synchronize()
            best[name] = min(best[name], e0.elapsed_time(e1) / REP)
            for k in env:
                ops.dispatch_override(k, -1)
    for name, _ in arms:
        tot[name] += best[name] * mult
    print(f"{H:4d}^2 {Cin:5d}->{Cout:<5d} x{mult}      " + " ".join(f"{best[n]:7.3f}ms {flops / best[n] / 1e9:6.0f}TF {cfgs[n][3:]:>10s}" for n, _ in arms), flush=True)
print("per-step total (ms), incl. the slab reductions:      " + " ".join(f"{tot[n]:12.3f}" for n, _ in arms))
