"""Diagnostic (GPU box): the 2-D benchmark step (bs 32, 512², bf16) replayed as ONE hipGraph (graph.GraphedTrainStep: side-stream reductions folded onto the capture stream)
against the eager launches bench.py times - does the graph win anything when the queue never drains?    python scripts/graph_vs_eager.py [steps]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mdeical_image_segmentation_amd.engine2d import UNet2DEngine  # noqa: E402
from mdeical_image_segmentation_amd.graph import GraphedTrainStep  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 30
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(1)
x = torch.randn(32, 1, 512, 512, generator=g).to(dev)
t = torch.randint(0, 2, (32, 512, 512), generator=g).to(dev)


def timed(fn):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e3


eng = UNet2DEngine(1, 2, dtype=torch.bfloat16, device=dev, seed=0, lr=1e-5)


def eager():
    eng.forward(x, t, train=True)
    eng.backward()
    eng.optimizer_step()


res = []
for rnd in range(2):
    res.append(("eager", timed(eager)))
    eng2 = UNet2DEngine(1, 2, dtype=torch.bfloat16, device=dev, seed=0, lr=1e-5)
    gs = GraphedTrainStep(eng2, x, t)
    res.append(("graph", timed(lambda: gs())))
    gs.release()
    del gs, eng2
    torch.cuda.empty_cache()
for k, v in res:
    print(f"{k}: {v:.3f} ms/step = {32 / v * 1e3:.1f} img/s")
