"""Residual 3-D U-Net with squeeze & excitation (HIP, csrc/se3d.hip) against the golden from the reference's ResidualUNetSE3D
(tests/golden/g13_resunet_se3d.npz) and the fp64 CPU oracle; the scSE kernels alone against torch autograd."""
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


class _St:
    pass


@pytest.mark.parametrize("C,dtype", [(64, torch.float32), (128, torch.float32), (1024, torch.float32), (256, torch.bfloat16)])
def test_scse_kernels_against_torch(C, dtype):
    """y = max(e*cSE(e), e*sSE(e)) and its backward (with the ReLU mask of e fused) vs torch autograd in fp64"""
    from mdeical_image_segmentation_amd import ops
    gen = torch.Generator().manual_seed(C)
    N, D, H, W = 2, 3, 4, 5
    S = D * H * W
    e = F.relu(torch.randn(N, D, H, W, C, generator=gen)).to(dtype)
    W1, W2 = torch.randn(C, C, generator=gen) / C ** 0.5, torch.randn(C, C, generator=gen) / C ** 0.5
    b1, b2 = 0.1 * torch.randn(C, generator=gen), 0.1 * torch.randn(C, generator=gen)
    w, b0 = torch.randn(C, generator=gen) / C ** 0.5, 0.1 * torch.randn(1, generator=gen)
    g = torch.randn(N, D, H, W, C, generator=gen).to(dtype)
    pre = (e.double() + (e == 0) * -1.0).requires_grad_(True)          # a pre-activation whose ReLU is e
    ref_p = [t.double().requires_grad_(True) for t in (W1, b1, W2, b2, w, b0)]
    er = F.relu(pre)
    a = torch.sigmoid(F.linear(F.relu(F.linear(er.mean(dim=(1, 2, 3)), ref_p[0], ref_p[1])), ref_p[2], ref_p[3]))
    bg = torch.sigmoid((er * ref_p[4]).sum(-1, keepdim=True) + ref_p[5])
    yr = torch.max(er * a[:, None, None, None, :], er * bg)
    yr.backward(g.double())
    st = _St()
    for name in ("sum", "sq", "mean", "z1", "a", "da", "cross"):
        setattr(st, name, torch.zeros(N, C, device=DEV))
    st.bgate, st.dq = torch.zeros(N, S, device=DEV), torch.zeros(N, S, device=DEV)
    dev = [t.to(DEV) for t in (W1, b1, W2, b2, w, b0)]
    ed, y = e.to(DEV), torch.empty(N, D, H, W, C, dtype=dtype, device=DEV)
    ops.se_fwd(ed, y, *dev, st)
    tol = 2e-5 if dtype == torch.float32 else 2e-2

    def rel(x, r):
        return ((x.detach().double().cpu() - r).norm() / r.norm().clamp_min(1e-30)).item()

    assert rel(y, yr.detach()) < tol
    gd = g.to(DEV).clone()
    grads = [torch.empty_like(t) for t in dev]
    ops.se_bwd(gd, ed, dev[0], dev[2], dev[4], st, *grads)
    assert rel(gd, pre.grad) < 5 * tol, "dL/d(pre-activation)"
    for name, got, r in zip(("dW1", "db1", "dW2", "db2", "dw", "db0"), grads, ref_p):
        assert rel(got, r.grad) < 10 * tol, name


def test_fp32_resunet_se3d_engine_matches_reference_golden_and_fp64_oracle():
    from mdeical_image_segmentation_amd.engine3d_res import ResidualUNetSE3DEngine
    from oracle import unet3d_oracle as o3
    g = load_golden("g13_resunet_se3d.npz")
    eng = ResidualUNetSE3DEngine(1, 3, f_maps=(64, 128, 256), dtype=torch.float32, device=DEV, seed=0)
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
    assert torch.allclose(eng.Gr["final_conv.weight"].cpu(), T(g["g_final_w"]), rtol=2e-3, atol=1e-6)
    p = {n: eng.P[n].detach().cpu().clone() for n in names}

    def grads(dt):
        ps_ = {k: v.detach().clone().to(dt).requires_grad_(True) for k, v in p.items()}
        lg = o3.resunet3d_forward(ps_, x.to(dt), 3)
        o3.bce_dice_loss(lg, t.to(dt)).backward()
        return lg.detach(), {k: v.grad for k, v in ps_.items()}

    (lg32, g32), (_, g64) = grads(torch.float32), grads(torch.float64)
    assert (lg32 - T(g["logits"])).abs().max().item() < 1e-5, "the oracle restates the reference module"
    worst = 0.0
    for n in names:
        nrm = g64[n].norm().item() + 1e-30
        err = (eng.Gr[n].cpu().double() - g64[n]).norm().item() / nrm
        ref_err = (g32[n].double() - g64[n]).norm().item() / nrm
        worst = max(worst, err / max(ref_err, 1e-9))
        if n.startswith("encoders.0") or n.startswith("encoders.1"):
            # upstream of ONE ReLU mask flip (traced): a pre-activation of encoder 1's block output is > 0 in this engine's fp32 summation order and
            # <= 0 in fp64; that voxel carries |g| = 2.4e-5, 3x the typical magnitude, = 1.7e-2 of the block gradient's norm.  Everything that
            # enters the block backward (pooled / joined gradients, arg-max routing, the scSE kernels on the oracle's gradient) agrees to 1e-6.
            assert err <= 5e-2, (n, err, ref_err)
        else:
            assert err <= max(4 * ref_err, 5e-4), (n, err, ref_err)
    print(f"residual SE 3-D: logits max|diff| {d:.3g}; worst (engine err / reference-fp32 err) vs fp64, rel. L2: {worst:.2f}")
    eng.optimizer_step()
    assert np.isfinite(eng.gradnorm.item())


def test_bf16_resunet_se3d_engine_close_and_mirror_module():
    from mdeical_image_segmentation_amd.engine3d_res import ResidualUNetSE3DEngine
    from mdeical_image_segmentation_amd.model.unet3d.losses import get_loss_criterion
    from mdeical_image_segmentation_amd.model.unet3d.model import get_model
    g = load_golden("g13_resunet_se3d.npz")
    eng = ResidualUNetSE3DEngine(1, 3, f_maps=(64, 128, 256), dtype=torch.bfloat16, device=DEV, seed=0)
    loss, logits, _ = eng.forward(T(g["x"]).to(DEV), T(g["t"]).to(DEV), train=True)
    eng.backward()
    ref = T(g["logits"])
    rel = (logits.cpu() - ref).abs().max().item() / ref.abs().max().item()
    assert rel < 0.08 and abs(loss.item() - float(g["loss"])) < 3e-2, (rel, loss.item())
    torch.manual_seed(0)
    m = get_model({"name": "ResidualUNetSE3D", "in_channels": 1, "out_channels": 3, "f_maps": [64, 128, 256], "num_levels": 3}).cuda()
    assert [k for k, _ in m.named_parameters()] == [str(n) for n in g["names"]]
    out = m(T(g["x"]).cuda())
    assert (out.detach().cpu() - ref).abs().max().item() < 1e-4
    crit = get_loss_criterion({"loss": {"name": "BCEDiceLoss"}})
    crit(out, T(g["t"]).cuda()).backward()
    assert torch.allclose(m.final_conv.weight.grad.cpu(), T(g["g_final_w"]), rtol=2e-3, atol=1e-6)
    assert all(p.grad is not None and torch.isfinite(p.grad).all() for p in m.parameters())
