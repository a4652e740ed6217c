"""G17: `create_conv` layer orders other than 'gcr', from the REAL reference modules (model/unet3d/buildingblocks.py:14-159 SingleConv, :255-325 ResNetBlock,
model/unet3d/model.py:197-232 ResidualUNet3D with its default order 'cge'), plus the data-derived / channel-wise options of Standardize and Normalize
(augment/unet3d_augment/transforms.py:495-523, 547-605).  Round 2 checked these routes against a float64 copy of the MIRROR's own module tree:
if the mirror mis-built an order, both sides agreed and were wrong.  Here the module trees are the reference's.

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_orders.py

Per case: the seeded input, the full state dict (so that the mirror is loaded with the reference's parameters, whatever its own init does), the output, the gradient
w.r.t. the input and every parameter gradient; BatchNorm cases also store the running statistics after the training-mode forward and an eval-mode output."""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from _ref_import import import_reference  # noqa: E402

torch.set_num_threads(8)
torch.use_deterministic_algorithms(True)

# (key, order, cin, cout, (N, D, H, W)): SingleConv(cin, cout, order=order) with the reference's defaults (kernel 3, padding 1, 8 groups)
SINGLE = [("cge", "cge", 24, 40, (2, 5, 6, 7)), ("cl", "cl", 8, 16, (1, 4, 6, 5)), ("crg", "crg", 16, 32, (2, 4, 5, 6)), ("gcl", "gcl", 16, 24, (1, 4, 6, 6)),
          ("bcr", "bcr", 12, 20, (3, 4, 5, 6)), ("cbr", "cbr", 12, 20, (3, 4, 5, 6)), ("cbl", "cbl", 6, 10, (2, 3, 5, 5)),
          ("gcrd", "gcrd", 16, 16, (1, 4, 5, 6)), ("cbrD", "cbrD", 8, 12, (2, 3, 4, 5))]


def run(module, x, gy, out, key, bn=False):
    x = x.clone().requires_grad_(True)
    module.train()
    y = module(x)
    y.backward(gy)
    out[f"{key}/x"], out[f"{key}/gy"], out[f"{key}/y"], out[f"{key}/dx"] = x.detach().numpy(), gy.numpy(), y.detach().numpy(), x.grad.numpy()
    names = []
    for n, p in module.named_parameters():
        names.append(n)
        out[f"{key}/p/{n}"] = p.detach().numpy().copy()
        out[f"{key}/g/{n}"] = p.grad.numpy().copy()
    out[f"{key}/names"] = np.array(names)
    for n, b in module.named_buffers():            # running statistics AFTER the training forward (num_batches_tracked included)
        out[f"{key}/b/{n}"] = b.detach().numpy().copy()
    if bn:
        module.eval()
        with torch.no_grad():
            out[f"{key}/y_eval"] = module(x.detach()).numpy()


def main():
    ns = import_reference()
    bb = ns.bb3d
    out = {}
    g = torch.Generator().manual_seed(1701)
    for key, order, cin, cout, (N, D, H, W) in SINGLE:
        torch.manual_seed(100 + len(out))
        m = bb.SingleConv(cin, cout, order=order, dropout_prob=0.25)
        with torch.no_grad():                       # non-trivial affine parameters
            for n, p in m.named_parameters():
                if "groupnorm" in n or "batchnorm" in n:
                    p.add_(0.3 * torch.randn(p.shape, generator=g))
        x = torch.randn(N, cin, D, H, W, generator=g) * 1.5 + 0.3
        gy = torch.randn(N, cout, D, H, W, generator=g)
        if "d" in order.lower():
            m.eval()                                # dropout: the deterministic (eval) behaviour is pinned; training-mode dropout is a random stream
            x2 = x.clone().requires_grad_(True)
            y = m(x2)
            y.backward(gy)
            out[f"{key}/x"], out[f"{key}/gy"], out[f"{key}/y_eval"], out[f"{key}/dx_eval"] = x.numpy(), gy.numpy(), y.detach().numpy(), x2.grad.numpy()
            out[f"{key}/names"] = np.array([n for n, _ in m.named_parameters()])
            for n, p in m.named_parameters():
                out[f"{key}/p/{n}"] = p.detach().numpy().copy()
                out[f"{key}/g/{n}"] = p.grad.numpy().copy()
            for n, b in m.named_buffers():
                out[f"{key}/b/{n}"] = b.detach().numpy().copy()
            continue
        run(m, x, gy, out, key, bn="b" in order)
    # ResNetBlock with the reference's default order 'cge' and with 'cle' .. 'gcl' variants
    for key, order, cin, cout in (("res_cge", "cge", 12, 24), ("res_gcl", "gcl", 16, 16)):
        torch.manual_seed(7)
        m = bb.ResNetBlock(cin, cout, order=order)
        x = torch.randn(2, cin, 4, 6, 5, generator=g)
        gy = torch.randn(2, cout, 4, 6, 5, generator=g)
        run(m, x, gy, out, key)
    # the whole residual net with ITS default order ('gcr' is what model.py passes; 'cge' is ResNetBlock's own default): ResidualUNet3D(layer_order='cge')
    torch.manual_seed(3)
    net = ns.model3d.ResidualUNet3D(1, 2, f_maps=[8, 16, 32], num_levels=3, layer_order="cge")
    x = torch.randn(1, 1, 8, 12, 8, generator=g)
    gy = torch.randn(1, 2, 8, 12, 8, generator=g)
    run(net, x, gy, out, "resunet_cge")
    # augment options
    tr = ns.transforms
    rng = np.random.RandomState(11)
    c4 = (rng.rand(3, 4, 6, 8).astype(np.float32) * np.array([1.0, 3.0, 0.2], dtype=np.float32).reshape(3, 1, 1, 1) + np.array([0.0, -2.0, 5.0], dtype=np.float32).reshape(3, 1, 1, 1))
    v = (rng.rand(6, 10, 12).astype(np.float32) * 7 - 2)
    out["aug/c4"], out["aug/v"] = c4, v
    out["aug/std_channelwise"] = tr.Standardize(channelwise=True)(c4)
    out["aug/norm_data"] = tr.Normalize()(v)
    out["aug/norm_data01"] = tr.Normalize(norm01=True)(v)
    out["aug/norm_min_only"] = tr.Normalize(min_value=-1.0)(v)
    out["aug/norm_channelwise"] = tr.Normalize(channelwise=True)(c4)
    out["aug/norm_channelwise_mixed"] = tr.Normalize(min_value=["None", -2.5, 5.0], channelwise=True)(c4)     # (both lists at once trip the reference's own assert)
    res = {k: np.asarray(a) for k, a in out.items()}
    np.savez_compressed(os.path.join(HERE, "g17_orders.npz"), **res)
    print("wrote g17_orders.npz", sum(a.nbytes for a in res.values()) // 1024, "KiB,", len(res), "arrays")


if __name__ == "__main__":
    main()
