"""G13: the reference's ResidualUNetSE3D (model/unet3d/model.py:235-280: ResNetBlockSE = ResNetBlock + scSE squeeze & excitation).

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_resunet_se3d.py

f_maps 64-128-256 (3 levels) on 1x1x16^3: logits / loss in full, parameters and gradients as statistics."""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from _ref_import import import_reference  # noqa: E402
from make_golden import stat  # noqa: E402

torch.set_num_threads(8)
torch.use_deterministic_algorithms(True)


def main():
    ns = import_reference()
    crit = ns.losses3d.BCEDiceLoss(1.0, 1.0)
    torch.manual_seed(0)
    net = ns.model3d.ResidualUNetSE3D(1, 3, f_maps=[64, 128, 256], num_levels=3)
    g = torch.Generator().manual_seed(83)
    x = torch.randn(1, 1, 16, 16, 16, generator=g)
    t = (torch.rand(1, 3, 16, 16, 16, generator=g) > 0.5).float()
    logits = net(x)
    loss = crit(logits, t)
    loss.backward()
    d = {"x": x, "t": t, "logits": logits, "loss": loss,
         "names": np.array([k for k, _ in net.named_parameters()]),
         "shapes": np.array([str(tuple(p.shape)) for _, p in net.named_parameters()]),
         "param_stats": np.stack([stat(p) for _, p in net.named_parameters()]),
         "grad_stats": np.stack([stat(p.grad) for _, p in net.named_parameters()]),
         "g_final_w": net.final_conv.weight.grad}
    out = {k: (v.detach().cpu().numpy() if isinstance(v, torch.Tensor) else np.asarray(v)) for k, v in d.items()}
    np.savez_compressed(os.path.join(HERE, "g13_resunet_se3d.npz"), **out)
    print("wrote g13_resunet_se3d.npz", sum(a.nbytes for a in out.values()) // 1024, "KiB")
    for n, s in zip(out["names"], out["shapes"]):
        print(" ", n, s)


if __name__ == "__main__":
    main()
