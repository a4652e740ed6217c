"""Residual 3-D U-Net engine (HIP) against the golden from the reference's ResidualUNet3D (tests/golden/g10_resunet3d.npz) and the fp64 CPU
oracle; the small kernels (residual add, single-channel 1x1x1 conv) against torch."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from conftest import load_golden

pytestmark = pytest.mark.gpu
DEV = "cuda"


def T(a):
    return torch.from_numpy(np.asarray(a))


def stat(t):
    t = t.detach().double().cpu().flatten()
    idx = torch.linspace(0, t.numel() - 1, steps=64).long()
    return np.concatenate([[t.sum().item(), t.abs().sum().item(), (t * t).sum().item()], t[idx].numpy()])


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_add_act_and_expand1(dtype):
    from mdeical_image_segmentation_amd import ops
    gen = torch.Generator().manual_seed(2)
    a = torch.randn(2, 3, 4, 5, 64, generator=gen).to(dtype)
    b = torch.randn(2, 3, 4, 5, 64, generator=gen).to(dtype)
    y = torch.empty(2, 3, 4, 5, 64, dtype=dtype, device=DEV)
    ops.add_act(a.to(DEV), b.to(DEV), y, relu=True)
    assert torch.equal(y.cpu(), F.relu(a.float() + b.float()).to(dtype))
    x = torch.randn(2, 1, 3, 4, 5, generator=gen)
    w, bias = torch.randn(64, generator=gen), torch.randn(64, generator=gen)
    out = torch.empty(2, 3, 4, 5, 64, dtype=dtype, device=DEV)
    ops.expand1_fwd(x.to(DEV), w.to(DEV), bias.to(DEV), out)
    ref = F.conv3d(x, w.view(64, 1, 1, 1, 1), bias).permute(0, 2, 3, 4, 1)
    tol = 1e-6 if dtype == torch.float32 else 3e-2
    assert (out.float().cpu() - ref).abs().max().item() < tol
    dy = torch.randn(2, 3, 4, 5, 64, generator=gen).to(dtype)
    dw, db = torch.empty(64, device=DEV), torch.empty(64, device=DEV)
    ops.expand1_bwd(x.to(DEV), dy.to(DEV), dw, db)
    assert torch.allclose(dw.cpu(), (x.permute(0, 2, 3, 4, 1) * dy.float()).sum((0, 1, 2, 3)), rtol=1e-4, atol=1e-4)
    assert torch.allclose(db.cpu(), dy.float().sum((0, 1, 2, 3)), rtol=1e-4, atol=1e-4)


def test_fp32_resunet3d_engine_matches_reference_golden_and_fp64_oracle():
    from mdeical_image_segmentation_amd.engine3d_res import ResidualUNet3DEngine
    from oracle import unet3d_oracle as o3
    g = load_golden("g10_resunet3d.npz")
    eng = ResidualUNet3DEngine(1, 3, f_maps=(64, 128, 256), dtype=torch.float32, device=DEV, seed=0)
    names = [str(n) for n in g["names"]]
    assert [n for n, _ in eng.specs] == names
    ps = np.stack([stat(eng.P[n]) for n in names])
    assert np.array_equal(ps[:, 3:], g["param_stats"][:, 3:]), "seeded init differs from the reference"
    x, t = T(g["x"]), T(g["t"])
    loss, logits, _ = eng.forward(x.to(DEV), t.to(DEV), train=True)
    d = (logits.cpu() - T(g["logits"])).abs().max().item()
    assert d < 1e-4, f"logits max|diff| {d}"
    assert abs(loss.item() - float(g["loss"])) < 1e-4
    eng.backward()
    gs = np.stack([stat(eng.Gr[n]) for n in names])
    refg = g["grad_stats"]
    for i, n in enumerate(names):
        assert abs(gs[i, 1] - refg[i, 1]) <= 5e-3 * abs(refg[i, 1]) + 1e-6, (n, "abssum", gs[i, 1], refg[i, 1])
    assert torch.allclose(eng.Gr["final_conv.weight"].cpu(), T(g["g_final_w"]), rtol=2e-3, atol=1e-6)
    p = {n: eng.P[n].detach().cpu().clone() for n in names}

    def grads(dt):
        ps_ = {k: v.detach().clone().to(dt).requires_grad_(True) for k, v in p.items()}
        lg = o3.resunet3d_forward(ps_, x.to(dt), 3)
        o3.bce_dice_loss(lg, t.to(dt)).backward()
        return {k: v.grad for k, v in ps_.items()}

    g32, g64 = grads(torch.float32), grads(torch.float64)
    worst = 0.0
    for n in names:
        nrm = g64[n].norm().item() + 1e-30
        err = (eng.Gr[n].cpu().double() - g64[n]).norm().item() / nrm
        ref_err = (g32[n].double() - g64[n]).norm().item() / nrm
        worst = max(worst, err / max(ref_err, 1e-9))
        assert err <= max(4 * ref_err, 3e-3), (n, err, ref_err)
    print(f"residual 3-D: logits max|diff| {d:.3g}; worst (engine err / reference-fp32 err) vs fp64, rel. L2: {worst:.2f}")
    eng.optimizer_step()
    assert np.isfinite(eng.gradnorm.item())


def test_bf16_resunet3d_engine_close():
    from mdeical_image_segmentation_amd.engine3d_res import ResidualUNet3DEngine
    g = load_golden("g10_resunet3d.npz")
    eng = ResidualUNet3DEngine(1, 3, f_maps=(64, 128, 256), dtype=torch.bfloat16, device=DEV, seed=0)
    loss, logits, _ = eng.forward(T(g["x"]).to(DEV), T(g["t"]).to(DEV), train=True)
    eng.backward()
    ref = T(g["logits"])
    rel = (logits.cpu() - ref).abs().max().item() / ref.abs().max().item()
    assert rel < 0.08 and abs(loss.item() - float(g["loss"])) < 3e-2, (rel, loss.item())
    a, b = eng.Gr["final_conv.weight"].cpu().flatten(), T(g["g_final_w"]).flatten()
    assert torch.dot(a, b) / (a.norm() * b.norm()) > 0.98


def test_residual_unet3d_mirror_module_autograd():
    """model.unet3d.model.ResidualUNet3D (nn.Module mirror, reference constructor arguments) -> logits; external BCEDiceLoss; fused backward"""
    from mdeical_image_segmentation_amd.model.unet3d.losses import get_loss_criterion
    from mdeical_image_segmentation_amd.model.unet3d.model import ResidualUNet3D, get_model
    g = load_golden("g10_resunet3d.npz")
    torch.manual_seed(0)
    m = ResidualUNet3D(1, 3, f_maps=[64, 128, 256], num_levels=3).cuda()
    assert [k for k, _ in m.named_parameters()] == [str(n) for n in g["names"]]
    logits = m(T(g["x"]).cuda())
    assert (logits.detach().cpu() - T(g["logits"])).abs().max().item() < 1e-4
    loss = get_loss_criterion({"loss": {"name": "BCEDiceLoss", "alpha": 1.0, "beta": 1.0}})(logits, T(g["t"]).cuda())
    assert abs(loss.item() - float(g["loss"])) < 1e-4
    loss.backward()
    refg = g["grad_stats"]
    for i, (k, p) in enumerate(m.named_parameters()):
        a = p.grad.detach().double().cpu().flatten()
        assert abs(a.abs().sum().item() - refg[i, 1]) <= 5e-3 * abs(refg[i, 1]) + 1e-6, k
    assert isinstance(get_model({"name": "ResidualUNet3D", "in_channels": 1, "out_channels": 3, "f_maps": [64, 128], "num_levels": 2}), ResidualUNet3D)
