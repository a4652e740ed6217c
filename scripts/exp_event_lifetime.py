"""Determinism of the side-stream slab reductions: K back-to-back UNSYNCHRONISED train steps from the same seeded state, repeated; the final
parameters and gradients must be bit-identical.  The 2-D engine puts its reductions on the side stream only with MISAMD_SIDE_REDUCE=1 (set here); the 3-D engines do
by default.  The ordering events are a process-lifetime ring per device (csrc/wgrad.hip wg_finish); tests/test_gpu_fullsize.py holds the same statement as a test.
    python scripts/exp_event_lifetime.py"""
import os
import sys

os.environ["MISAMD_SIDE_REDUCE"] = "1"
import torch  # noqa: E402

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mdeical_image_segmentation_amd.engine2d import UNet2DEngine  # noqa: E402
from mdeical_image_segmentation_amd.engine3d import UNet3DEngine  # noqa: E402

K, REPS = 8, 4
gen = torch.Generator().manual_seed(3)
ok = True
for kind in ("2d", "3d"):
    if kind == "2d":
        x = torch.randn(16, 1, 256, 256, generator=gen).cuda()
        y = torch.randint(0, 2, (16, 256, 256), generator=gen).cuda()
    else:
        x = torch.randn(2, 1, 64, 64, 64, generator=gen).cuda()
        y = (torch.rand(2, 3, 64, 64, 64, generator=gen) > 0.5).float().cuda()
    finals = []
    for rep in range(REPS):
        eng = (UNet2DEngine(1, 2, dtype=torch.bfloat16, device="cuda", seed=0, lr=1e-4) if kind == "2d" else
               UNet3DEngine(1, 3, dtype=torch.bfloat16, device="cuda", seed=0, lr=1e-4))
        assert eng.side_reduce
        for _ in range(K):
            eng.train_step(x, y)
        torch.cuda.synchronize()
        finals.append((eng.flat.p.clone(), eng.flat.g.clone()))
    same = all(torch.equal(finals[0][0], f[0]) and torch.equal(finals[0][1], f[1]) for f in finals[1:])
    ok = ok and same
    print(f"[{kind}, side-stream reductions] {REPS} x {K} unsynchronised steps: final parameters and gradients bit-identical across repetitions: {same}")
sys.exit(0 if ok else 1)
