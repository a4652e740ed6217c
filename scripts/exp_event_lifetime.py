"""Experiment for ADVICE r1 / VERDICT r1 weak #12: is the process-lifetime event ring in wgrad.hip needed?
Runs K back-to-back (unsynchronised) train steps from the same seeded state twice and compares the final parameters bit for bit.
    python scripts/exp_event_lifetime.py                 # event ring
    MIS_WGRAD_EVENT_PER_CALL=1 python scripts/exp_event_lifetime.py   # create / record / wait / destroy per call"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mdeical_image_segmentation_amd.engine2d import UNet2DEngine  # noqa: E402

B, S, K, REPS = 16, 256, 8, 4
gen = torch.Generator().manual_seed(3)
x = torch.randn(B, 1, S, S, generator=gen).cuda()
y = torch.randint(0, 2, (B, S, S), generator=gen).cuda()
finals = []
for rep in range(REPS):
    eng = UNet2DEngine(1, 2, dtype=torch.bfloat16, device="cuda", seed=0, lr=1e-4)
    for _ in range(K):
        eng.train_step(x, y)
    torch.cuda.synchronize()
    finals.append((eng.flat.p.clone(), eng.flat.g.clone()))
same = all(torch.equal(finals[0][0], f[0]) and torch.equal(finals[0][1], f[1]) for f in finals[1:])
mode = "per-call events" if os.environ.get("MIS_WGRAD_EVENT_PER_CALL") else "event ring"
print(f"[{mode}] {REPS} x {K} unsynchronised steps: final parameters and gradients bit-identical across repetitions: {same}")
sys.exit(0 if same else 1)
