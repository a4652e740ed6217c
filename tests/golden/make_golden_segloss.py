"""G19: the reference's OWN loss terms of SegmentationLoss - F1Loss and IoULoss (model/unet2d/loss.py:32-56) - on seeded logits / targets, CPU, build container only.

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_segloss.py

The third term (MSSSIMLoss) is pytorch_msssim 1.0.0, a third-party package absent from this image: it stays "parity unpinned" (oracle/segloss_oracle.py
restates its published algorithm).  The two terms the reference defines itself are pinned here: value and dL/dlogits of each, two shapes (one odd), logits with a wide
range so that sigmoid saturation is exercised."""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from _ref_import import import_reference  # noqa: E402

torch.set_num_threads(8)


def main():
    import_reference()
    import model.unet2d.loss as RL          # the reference module (pytorch_msssim is a stand-in: MSSSIMLoss / SegmentationLoss are not touched)
    out = {}
    for i, shape in enumerate([(2, 1, 64, 80), (1, 1, 53, 41), (3, 1, 33, 47)]):
        gen = torch.Generator().manual_seed(190 + i)
        t = (torch.rand(*shape, generator=gen) > 0.6).float()
        x = torch.randn(*shape, generator=gen) * 2.5 + (t * 2 - 1)
        out[f"x{i}"], out[f"t{i}"] = x.numpy(), t.numpy().astype(np.uint8)
        for name, cls in (("f1", RL.F1Loss), ("iou", RL.IoULoss)):
            xi = x.clone().requires_grad_(True)
            loss = cls()(xi, t)
            loss.backward()
            out[f"{name}{i}"] = np.float64(loss.item())
            out[f"{name}{i}_grad"] = xi.grad.numpy()
            print(name, shape, loss.item(), float(xi.grad.abs().max()))
    np.savez_compressed(os.path.join(HERE, "g19_segloss.npz"), **out)
    print(sum(a.nbytes for a in out.values()) // 1024, "KiB")


if __name__ == "__main__":
    main()
