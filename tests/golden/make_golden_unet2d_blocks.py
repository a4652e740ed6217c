"""G18: the is3d=False variants of the 3-D package's blocks and networks, from the REAL reference modules (model/unet3d/buildingblocks.py: create_conv with Conv2d /
BatchNorm2d :65-104, Encoder with MaxPool2d :409-418, TransposeConvUpsampling with ConvTranspose2d :700-718; model/unet3d/model.py:283-359 UNet2D, ResidualUNet2D).

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_unet2d_blocks.py

Per case: the seeded input, the full state dict, the output, the gradient w.r.t. the input and every parameter gradient (training mode; the orders used have no dropout)."""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from _ref_import import import_reference  # noqa: E402
from make_golden_orders import run  # noqa: E402

torch.set_num_threads(8)
torch.use_deterministic_algorithms(True)


def main():
    ns = import_reference()
    bb, md = ns.bb3d, ns.model3d
    out = {}
    g = torch.Generator().manual_seed(1801)

    def jitter(m):
        with torch.no_grad():
            for n, p in m.named_parameters():
                if "groupnorm" in n or "batchnorm" in n:
                    p.add_(0.3 * torch.randn(p.shape, generator=g))

    # SingleConv 2-D: GroupNorm in front ('gcr'), BatchNorm2d behind ('cbr')
    for key, order, cin, cout, (N, H, W) in (("sc_gcr", "gcr", 16, 24, (2, 9, 11)), ("sc_cbr", "cbr", 12, 20, (3, 8, 10))):
        torch.manual_seed(50 + len(out))
        m = bb.SingleConv(cin, cout, order=order, is3d=False)
        jitter(m)
        run(m, torch.randn(N, cin, H, W, generator=g) * 1.5 + 0.3, torch.randn(N, cout, H, W, generator=g), out, key, bn="b" in order)
    # Encoder 2-D with MaxPool2d(2) in front of a DoubleConv
    torch.manual_seed(61)
    m = bb.Encoder(8, 16, basic_module=bb.DoubleConv, conv_layer_order="gcr", num_groups=4, is3d=False)
    jitter(m)
    run(m, torch.randn(2, 8, 12, 10, generator=g), torch.randn(2, 16, 6, 5, generator=g), out, "enc2d")
    # ResNetBlock 2-D (1x1 Conv2d on the residual path)
    torch.manual_seed(62)
    m = bb.ResNetBlock(12, 24, order="cge", is3d=False)
    jitter(m)
    run(m, torch.randn(2, 12, 7, 9, generator=g), torch.randn(2, 24, 7, 9, generator=g), out, "res2d")
    # the networks
    torch.manual_seed(63)
    net = md.UNet2D(1, 2, f_maps=[8, 16, 32], num_groups=4, num_levels=3)
    jitter(net)
    run(net, torch.randn(2, 1, 16, 24, generator=g), torch.randn(2, 2, 16, 24, generator=g), out, "unet2d")
    torch.manual_seed(64)
    net = md.ResidualUNet2D(1, 2, f_maps=[8, 16, 32], num_groups=4, num_levels=3)
    jitter(net)
    run(net, torch.randn(1, 1, 16, 16, generator=g), torch.randn(1, 2, 16, 16, generator=g), out, "resunet2d")
    path = os.path.join(HERE, "g18_unet2d_blocks.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes;", sorted({k.split('/')[0] for k in out}))


if __name__ == "__main__":
    main()
