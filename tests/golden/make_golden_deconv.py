"""G3d: the reference's UNet3D with upsample='deconv' (TransposeConvUpsampling, model/unet3d/buildingblocks.py:676-728).

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_deconv.py

(a) a small net stored in full (f_maps [8, 16, 32]; the transposed conv yields 2n-1 voxels per axis, the nearest resize 2n),
(b) the engine-sized net (f_maps 64..256, 3 levels) on 1x1x16^3: logits / loss in full, gradients as statistics.
"""
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
    from oracle import unet3d_oracle as o3
    crit = ns.losses3d.BCEDiceLoss(1.0, 1.0)
    d = {}
    torch.manual_seed(0)
    net = ns.model3d.UNet3D(1, 3, f_maps=[8, 16, 32], num_groups=4, upsample="deconv")
    po = o3.init_params(1, 3, f_maps=[8, 16, 32], seed=0, upsample="deconv")
    sd = net.state_dict()
    assert list(sd.keys()) == list(po.keys()), (list(sd.keys()), list(po.keys()))
    for k in sd:
        assert torch.equal(sd[k], po[k]), k
    g = torch.Generator().manual_seed(79)
    x = torch.randn(2, 1, 8, 12, 20, generator=g)
    t = (torch.rand(2, 3, 8, 12, 20, generator=g) > 0.5).float()
    logits = net(x)
    loss = crit(logits, t)
    loss.backward()
    d.update({"s_x": x, "s_t": t, "s_logits": logits, "s_loss": loss})
    for k, v in net.named_parameters():
        d["s_p_" + k] = v
        d["s_g_" + k] = v.grad
    torch.manual_seed(0)
    fm = [64, 128, 256]
    net = ns.model3d.UNet3D(1, 3, f_maps=fm, num_levels=3, upsample="deconv")
    g = torch.Generator().manual_seed(80)
    x = torch.randn(1, 1, 16, 16, 16, generator=g)
    t = (torch.rand(1, 3, 16, 16, 16, generator=g) > 0.5).float()
    logits = net(x)
    loss = crit(logits, t)
    loss.backward()
    d.update({"x": x, "t": t, "logits": logits, "loss": loss,
              "names": np.array([k for k, _ in net.named_parameters()]),
              "param_stats": np.stack([stat(p) for _, p in net.named_parameters()]),
              "grad_stats": np.stack([stat(p.grad) for _, p in net.named_parameters()]),
              "g_final_w": net.final_conv.weight.grad})
    out = {k: (v.detach().cpu().numpy() if isinstance(v, torch.Tensor) else np.asarray(v)) for k, v in d.items()}
    np.savez_compressed(os.path.join(HERE, "g3_unet3d_deconv.npz"), **out)
    print("wrote g3_unet3d_deconv.npz", sum(a.nbytes for a in out.values()) // 1024, "KiB")


if __name__ == "__main__":
    main()
