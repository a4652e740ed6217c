"""G15: the reference's UNet_3Plus_DeepSup_CGM (model/unet2d/unet.py:795-1153), CPU, build container only.

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_cgm.py

Two train-mode forwards (BatchNorm running statistics move; the classifier's Dropout only changes the gate of outputs that are not stored), then the
eval-mode forward whose five gated probability maps, classifier scores and gates are stored, plus parameter names / seeded-init statistics."""
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


def main():
    import_reference()
    import model.unet2d.unet as U
    torch.manual_seed(5)
    net = U.UNet_3Plus_DeepSup_CGM(3, 1).train()
    with torch.no_grad():
        net.cls[1].bias.copy_(torch.tensor([0.05, 0.0]))      # after the seeded init: makes the weak inputs fall to class 0 (both gate values occur)
    out = {"names": np.array([k for k, _ in net.named_parameters()]), "state_keys": np.array(list(net.state_dict().keys())),
           "param_stats": np.stack([stat(p) for _, p in net.named_parameters()])}
    g = torch.Generator().manual_seed(51)
    xt = torch.randn(2, 2, 3, 32, 48, generator=g)
    xe = torch.randn(4, 3, 32, 48, generator=g) * torch.tensor([0.2, 1.0, 3.0, 0.05]).view(4, 1, 1, 1)
    with torch.no_grad():
        for b in xt:
            net(b)
        net.eval()
        cls = net.cls(net.conv5(net.maxpool4(net.conv4(net.maxpool3(net.conv3(net.maxpool2(net.conv2(net.maxpool1(net.conv1(xe))))))))))
        outs = net(xe)
    out.update({"xt": xt, "xe": xe, "cls": cls.squeeze(3).squeeze(2)})
    for i, o in enumerate(outs):
        out[f"d{i + 1}"] = o
    out = {k: (v.detach().cpu().numpy() if isinstance(v, torch.Tensor) else np.asarray(v)) for k, v in out.items()}
    np.savez_compressed(os.path.join(HERE, "g15_cgm.npz"), **out)
    print("cls", out["cls"], "gate", out["cls"].argmax(1), "d1 range", out["d1"].min(), out["d1"].max(), sum(a.nbytes for a in out.values()) // 1024, "KiB")


if __name__ == "__main__":
    main()
