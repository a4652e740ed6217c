"""time the weight repack of the fused engines in isolation (GPU box): python scripts/time_repack.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mdeical_image_segmentation_amd.engine2d import UNet2DEngine  # noqa: E402
from mdeical_image_segmentation_amd.engine3d import UNet3DEngine  # noqa: E402

for name, eng in (("2d", UNet2DEngine(1, 2, dtype=torch.bfloat16, device="cuda", seed=0)), ("3d", UNet3DEngine(1, 3, dtype=torch.bfloat16, device="cuda", seed=0))):
    eng.repack()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        eng.repack()
    e1.record()
    torch.cuda.synchronize()
    t = getattr(eng, "_pack_table", None)
    print(name, "repack %.3f ms" % (e0.elapsed_time(e1) / 10), "entries", getattr(t, "n", None), "max", getattr(t, "max_rows", None), getattr(t, "max_cols", None))
    if t:
        import numpy as np
        tab = t.dev.cpu().numpy().view(np.dtype([("w", "<u8"), ("wf", "<u8"), ("wd", "<u8"), ("rows", "<i4"), ("cols", "<i4"), ("taps", "<i4"), ("kind", "<i4")]))
        print([(int(r["rows"]), int(r["cols"]), int(r["taps"]), int(r["wd"] != 0)) for r in tab])
