"""G9: the reference's UNet_3Plus (model/unet2d/unet.py:136-446) in train mode on CPU.

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_unet3plus.py

Stores the input, logits, the gradient w.r.t. the input and per-parameter statistics of the seeded parameters and of their gradients
(27 M parameters: not stored, regenerated from the seed by the mirror; bit-identity of the seeded init is part of the test)."""
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
    import_reference()
    import model.unet2d.unet as U
    torch.manual_seed(3)
    net = U.UNet_3Plus(3, 1).train()
    g = torch.Generator().manual_seed(31)
    x = torch.randn(2, 3, 48, 80, generator=g, requires_grad=True)       # 48 = 3*16, 80 = 5*16: pooled grids 3x5 ... (ceil-mode windows clipped)
    names = [k for k, _ in net.named_parameters()]
    pstats = np.stack([stat(p) for _, p in net.named_parameters()])
    y = net(x)
    gy = torch.randn(y.shape, generator=g)
    y.backward(gy)
    sd = net.state_dict()
    out = {"x": x.detach(), "y": y.detach(), "gy": gy, "gx": x.grad, "names": np.array(names), "param_stats": pstats,
           "grad_stats": np.stack([stat(p.grad) for _, p in net.named_parameters()]),
           "state_keys": np.array(list(sd.keys())),
           "rm_conv1": sd["conv1.conv1.1.running_mean"], "rv_bn1d": sd["bn1d_1.running_var"],
           "g_outconv_w": net.outconv1.weight.grad, "g_outconv_b": net.outconv1.bias.grad, "g_h1cat_bn_w": net.h1_Cat_hd1_bn.weight.grad}
    # odd size: ceil-mode pooling windows are clipped (38 -> 19 -> 9 -> 4 -> 2; 8x pooling of 38 gives 5)
    net.eval()
    with torch.no_grad():
        xe = torch.randn(1, 3, 32, 32, generator=g)
        out["xe"], out["ye"] = xe, net(xe)
    # deep supervision (unet.py:454-787): five full-resolution logit maps
    torch.manual_seed(4)
    ds = U.UNet_3Plus_DeepSup(3, 1).train()
    xd = torch.randn(1, 3, 32, 48, generator=g)
    outs = ds(xd)
    sum(o.sum() * (i + 1) for i, o in enumerate(outs)).backward()
    out["ds_x"] = xd
    out["ds_names"] = np.array([k for k, _ in ds.named_parameters()])
    out["ds_state_keys"] = np.array(list(ds.state_dict().keys()))
    out["ds_param_stats"] = np.stack([stat(p) for _, p in ds.named_parameters()])
    for i, o in enumerate(outs):
        out[f"ds_d{i + 1}"] = o
    for k in range(1, 6):
        out[f"ds_g_outconv{k}_w"] = getattr(ds, f"outconv{k}").weight.grad
    out = {k: (v.detach().cpu().numpy() if isinstance(v, torch.Tensor) else np.asarray(v)) for k, v in out.items()}
    np.savez_compressed(os.path.join(HERE, "g9_unet3plus.npz"), **out)
    print("wrote g9_unet3plus.npz", sum(a.nbytes for a in out.values()) // 1024, "KiB;", len(names), "parameters")


if __name__ == "__main__":
    main()
