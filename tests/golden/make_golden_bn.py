"""G4: the reference's `unetConv2` (BatchNorm variant of the double conv, model/unet2d/layers.py:8-46) in train and eval mode.

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_bn.py

Runs the REAL reference module on CPU (seeded) and stores inputs, parameters, outputs, gradients and the running statistics.
"""
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


def main():
    ns = import_reference()
    L = ns.layers2d
    d = {}
    for tag, (cin, cout, n) in {"a": (3, 64, 2), "b": (64, 64, 1)}.items():
        torch.manual_seed(21)
        m = L.unetConv2(cin, cout, True, n=n)
        with torch.no_grad():     # non-trivial affine + running stats so that every term of the formulas is exercised
            for i in range(1, n + 1):
                bn = getattr(m, "conv%d" % i)[1]
                bn.bias.uniform_(-0.3, 0.3)
                bn.running_mean.uniform_(-0.2, 0.2)
                bn.running_var.uniform_(0.5, 1.5)
        for k, v in m.state_dict().items():
            d[f"{tag}_p0_{k}"] = v.clone()
        m.train()
        x = torch.randn(3, cin, 10, 12, requires_grad=True)
        y = m(x)
        gy = torch.randn_like(y)
        y.backward(gy)
        d.update({f"{tag}_x": x, f"{tag}_y": y, f"{tag}_gy": gy, f"{tag}_gx": x.grad})
        for k, v in m.named_parameters():
            d[f"{tag}_g_{k}"] = v.grad
        for k, v in m.state_dict().items():
            if "running" in k or "num_batches" in k:
                d[f"{tag}_p1_{k}"] = v.clone()
        m.eval()
        with torch.no_grad():
            d[f"{tag}_y_eval"] = m(x)
    # the no-norm variant (is_batchnorm=False) is a plain conv+ReLU chain
    torch.manual_seed(22)
    m = L.unetConv2(64, 64, False, n=2)
    x = torch.randn(2, 64, 8, 8)
    for k, v in m.state_dict().items():
        d[f"c_p0_{k}"] = v.clone()
    d.update({"c_x": x, "c_y": m(x)})
    out = {k: (v.detach().cpu().numpy() if isinstance(v, torch.Tensor) else np.asarray(v)) for k, v in d.items()}
    np.savez_compressed(os.path.join(HERE, "g4_unetconv2.npz"), **out)
    print("wrote g4_unetconv2.npz", sum(a.nbytes for a in out.values()) // 1024, "KiB")


if __name__ == "__main__":
    main()
