"""`unetConv2` (the reference's BatchNorm double conv, model/unet2d/layers.py:8-46) on the HIP path against the golden captured from
the real reference (tests/golden/g4_unetconv2.npz): train-mode output, all gradients, running statistics, eval-mode output."""
import numpy as np
import pytest
import torch

from conftest import load_golden

pytestmark = pytest.mark.gpu


def T(a):
    return torch.from_numpy(np.asarray(a))


def _load(m, g, tag):
    sd = {k[len(tag) + 4:]: T(g[k]) for k in g.files if k.startswith(f"{tag}_p0_")}
    assert set(sd) == set(m.state_dict()), "state-dict keys differ from the reference's"
    m.load_state_dict(sd)


@pytest.mark.parametrize("tag,cin,cout,n", [("a", 3, 64, 2), ("b", 64, 64, 1)])
def test_unetconv2_bn_train_and_eval(tag, cin, cout, n):
    from mdeical_image_segmentation_amd.model.unet2d.layers import unetConv2
    g = load_golden("g4_unetconv2.npz")
    m = unetConv2(cin, cout, True, n=n)
    _load(m, g, tag)
    m = m.cuda().train()
    x = T(g[f"{tag}_x"]).cuda().requires_grad_(True)
    y = m(x)
    d = (y.detach().cpu() - T(g[f"{tag}_y"])).abs().max().item()
    assert d < 2e-4, f"train-mode output max|diff| {d}"
    y.backward(T(g[f"{tag}_gy"]).cuda())
    ref_gx = T(g[f"{tag}_gx"])
    assert (x.grad.cpu() - ref_gx).abs().max().item() < 2e-4 * max(1.0, ref_gx.abs().max().item())
    for k, p in m.named_parameters():
        ref = T(g[f"{tag}_g_{k}"])
        err = (p.grad.cpu() - ref).abs().max().item()
        # conv biases in front of a BatchNorm have a mathematically zero gradient: pure rounding noise on both sides
        bound = 5e-4 if k.endswith("0.bias") else 5e-4 * max(1.0, ref.abs().max().item())
        assert err < bound, (k, err)
    sd = m.state_dict()
    for k in g.files:
        if k.startswith(f"{tag}_p1_"):
            name = k[len(tag) + 4:]
            assert torch.allclose(sd[name].cpu().double(), T(g[k]).double(), atol=1e-5), name
    m.eval()
    with torch.no_grad():
        ye = m(x.detach())
    de = (ye.cpu() - T(g[f"{tag}_y_eval"])).abs().max().item()
    assert de < 2e-4, f"eval-mode output max|diff| {de}"
    print(f"[{tag}] train max|diff| {d:.3g}, eval {de:.3g}")


def test_unetconv2_without_norm_and_init():
    from mdeical_image_segmentation_amd.model.unet2d.layers import unetConv2
    g = load_golden("g4_unetconv2.npz")
    m = unetConv2(64, 64, False, n=2)
    _load(m, g, "c")
    y = m.cuda()(T(g["c_x"]).cuda())
    assert (y.cpu() - T(g["c_y"])).abs().max().item() < 2e-4
    # same construction order / RNG consumption as the reference: kaiming-normal conv weights, N(1, 0.02) BN weights
    torch.manual_seed(21)
    m2 = unetConv2(3, 64, True, n=2)
    assert torch.equal(m2.conv1[0].weight, T(g["a_p0_conv1.0.weight"]))
    assert torch.equal(m2.conv2[1].weight, T(g["a_p0_conv2.1.weight"]))
    with pytest.raises(NotImplementedError):
        unetConv2(3, 64, True, ks=5)
