"""G20: the reference's squeeze & excitation layers called ON THEIR OWN (model/unet3d/se.py: ChannelSELayer3D :18-53, SpatialSELayer3D :56-98,
ChannelSpatialSELayer3D :101-116) and one ResNetBlockSE(se_module='cse') (model/unet3d/buildingblocks.py:326-362), CPU, build container only.

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_se_layers.py

Per case: seeded input (zero-mean, so both signs reach torch.max; case 3 is a ReLU output, so exact e == 0 ties reach its half / half rule), the layer's parameters,
the output, and the gradients of sum(y * r) with a seeded r with respect to the input and every parameter."""
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

CASES = [("cse", 64, 2, (2, 3, 4, 5), False), ("sse", 64, 2, (2, 3, 4, 5), False), ("scse", 64, 2, (2, 3, 4, 5), False), ("scse", 128, 4, (1, 2, 3, 5), True),
         ("cse", 48, 2, (1, 2, 3, 5), False), ("scse", 256, 8, (1, 2, 3, 3), False), ("scse", 64, 1, (1, 2, 3, 5), False)]


def main():
    ns = import_reference()
    import pytorch3dunet.unet3d.se as SE         # (= the reference's model/unet3d/se.py, loaded by _ref_import)
    bb = sys.modules[ns.model3d.__name__.rsplit(".", 1)[0] + ".buildingblocks"]
    out = {"cases": np.array([f"{k}:{C}:{r}:{'x'.join(map(str, g))}:{int(relu)}" for k, C, r, g, relu in CASES])}
    for i, (kind, C, r, grid, relu_in) in enumerate(CASES):
        torch.manual_seed(200 + i)
        layer = {"cse": lambda: SE.ChannelSELayer3D(C, r), "sse": lambda: SE.SpatialSELayer3D(C), "scse": lambda: SE.ChannelSpatialSELayer3D(C, r)}[kind]()
        gen = torch.Generator().manual_seed(300 + i)
        x = torch.randn(grid[0], C, *grid[1:], generator=gen)
        if relu_in:
            x = torch.relu(x)
        rr = torch.randn(grid[0], C, *grid[1:], generator=gen)
        xi = x.clone().requires_grad_(True)
        y = layer(xi)
        (y * rr).sum().backward()
        out[f"x{i}"], out[f"r{i}"], out[f"y{i}"], out[f"dx{i}"] = x.numpy(), rr.numpy(), y.detach().numpy(), xi.grad.numpy()
        for n, p in layer.named_parameters():
            out[f"p{i}.{n}"], out[f"g{i}.{n}"] = p.detach().numpy(), p.grad.numpy()
        print(kind, C, r, grid, float(y.detach().abs().mean()), float(xi.grad.abs().mean()))
    # one residual block with a cSE tail (the se_module values the fused network never passes)
    torch.manual_seed(260)
    blk = bb.ResNetBlockSE(32, 64, order="gcr", num_groups=8, se_module="cse")
    gen = torch.Generator().manual_seed(360)
    x = torch.randn(1, 32, 4, 4, 6, generator=gen)
    rr = torch.randn(1, 64, 4, 4, 6, generator=gen)
    xi = x.clone().requires_grad_(True)
    y = blk(xi)
    (y * rr).sum().backward()
    out["bx"], out["br"], out["by"], out["bdx"] = x.numpy(), rr.numpy(), y.detach().numpy(), xi.grad.numpy()
    # (the block's 2 x 64 x 64 x 27 weights would be 2 MB: parameters / gradients as statistics, as in g10 / g13; the test rebuilds the block under the same seed and
    # checks the parameter statistics first)
    out["bnames"] = np.array([n for n, _ in blk.named_parameters()])
    out["bparam_stats"] = np.stack([stat(p) for _, p in blk.named_parameters()])
    out["bgrad_stats"] = np.stack([stat(p.grad) for _, p in blk.named_parameters()])
    out["bg.se_module.fc1.weight"], out["bg.se_module.fc2.weight"] = blk.se_module.fc1.weight.grad.numpy(), blk.se_module.fc2.weight.grad.numpy()
    np.savez_compressed(os.path.join(HERE, "g20_se_layers.npz"), **out)
    print(sum(a.nbytes for a in out.values()) // 1024, "KiB")


if __name__ == "__main__":
    main()
