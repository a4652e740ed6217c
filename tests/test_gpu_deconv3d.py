"""upsample='deconv' (ConvTranspose3d k3 s2 p1 + nearest resize, reference model/unet3d/buildingblocks.py:676-728) on the HIP path:
the two gather kernels against torch, and the 3-level engine (f_maps 64-128-256) against the golden from the real reference and
the fp64 CPU oracle."""
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
def test_col2im_im2col_are_the_transposed_conv_scatter_and_its_adjoint(dtype):
    from mdeical_image_segmentation_amd import ops
    N, d, h, w, C = 2, 3, 4, 5, 16
    gen = torch.Generator().manual_seed(3)
    cols = torch.randn(N, d, h, w, 27 * C, generator=gen).to(dtype)
    # reference: scatter cols into the (2n-1) grid with F.fold-like loops on the CPU, then nearest resize to 2n
    cf = cols.float().view(N, d, h, w, 27, C)
    ct = torch.zeros(N, 2 * d - 1, 2 * h - 1, 2 * w - 1, C)
    for kd in range(3):
        for kh in range(3):
            for kw in range(3):
                for i in range(d):
                    for j in range(h):
                        for k in range(w):
                            pz, py, px = 2 * i - 1 + kd, 2 * j - 1 + kh, 2 * k - 1 + kw
                            if 0 <= pz < 2 * d - 1 and 0 <= py < 2 * h - 1 and 0 <= px < 2 * w - 1:
                                ct[:, pz, py, px] += cf[:, i, j, k, (kd * 3 + kh) * 3 + kw]
    ref_u = F.interpolate(ct.permute(0, 4, 1, 2, 3), size=(2 * d, 2 * h, 2 * w), mode="nearest").permute(0, 2, 3, 4, 1)
    u = torch.empty(N, 2 * d, 2 * h, 2 * w, C, dtype=dtype, device=DEV)
    ops.convt3_col2im(cols.to(DEV), u)
    tol = 1e-5 if dtype == torch.float32 else 4e-2
    assert (u.float().cpu() - ref_u).abs().max().item() < tol
    # adjoint: <col2im(cols), gu> == <cols, im2col(gu)>
    gu = torch.randn(N, 2 * d, 2 * h, 2 * w, C, generator=gen).to(dtype)
    gcols = torch.empty_like(cols, device=DEV)
    ops.convt3_im2col(gu.to(DEV), gcols)
    lhs = (ref_u.double() * gu.double()).sum().item()
    rhs = (cols.double() * gcols.cpu().double()).sum().item()
    assert abs(lhs - rhs) < (1e-4 if dtype == torch.float32 else 0.5) * max(1.0, abs(lhs)) * (1 if dtype == torch.float32 else 0.05)


def _engine(dtype):
    from mdeical_image_segmentation_amd.engine3d import UNet3DEngine
    return UNet3DEngine(1, 3, f_maps=(64, 128, 256), dtype=dtype, device=DEV, seed=0, upsample="deconv")


def test_fp32_deconv_engine_matches_reference_golden_and_fp64_oracle():
    from oracle import unet3d_oracle as o3
    g = load_golden("g3_unet3d_deconv.npz")
    eng = _engine(torch.float32)
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
        # single-element gradients (the 1-channel GroupNorm of encoders.0) are differences of large sums: fp32 noise on both sides
        slack = 2e-4 if eng.Gr[n].numel() == 1 else 1e-6
        assert abs(gs[i, 1] - refg[i, 1]) <= 3e-3 * abs(refg[i, 1]) + slack, (n, "abssum", gs[i, 1], refg[i, 1])
    assert torch.allclose(eng.Gr["final_conv.weight"].cpu(), T(g["g_final_w"]), rtol=2e-3, atol=1e-6)
    # every gradient tensor against the fp64 oracle in relative L2 (the reference's own fp32 backward is the noise yardstick, as in
    # test_gpu_engine3d).  L2 rather than max-norm: one ReLU pre-activation of this input sits at -2.7e-6 in fp64 and at +6e-7 in
    # fp32 arithmetic of a different summation order, which flips that voxel's mask and moves everything upstream by ~1.7e-3.
    p = o3.init_params(1, 3, f_maps=[64, 128, 256], num_levels=3, seed=0, upsample="deconv")
    _, _, g32 = o3.loss_and_grads(p, x, t, num_levels=3, upsample="deconv")
    _, _, g64 = o3.loss_and_grads({k: v.double() for k, v in p.items()}, x.double(), t.double(), num_levels=3, upsample="deconv")
    for n, gref in g64.items():
        nrm = gref.norm().item() + 1e-30
        err = (eng.Gr[n].cpu().double() - gref).norm().item() / nrm
        ref_err = (g32[n].double() - gref).norm().item() / nrm
        floor = 2e-4 / nrm if gref.numel() == 1 else 0.0
        assert err <= max(2 * ref_err, 3e-3) + floor, (n, err, ref_err)
        if n.startswith("decoders.1") or n.startswith("final"):      # downstream of no mask flip: tight
            assert err <= max(4 * ref_err, 2e-5), (n, err, ref_err)
    eng.optimizer_step()
    assert np.isfinite(eng.gradnorm.item())


def test_bf16_deconv_engine_close_and_mirror_model():
    g = load_golden("g3_unet3d_deconv.npz")
    eng = _engine(torch.bfloat16)
    loss, logits, _ = eng.forward(T(g["x"]).to(DEV), T(g["t"]).to(DEV), train=True)
    eng.backward()
    ref = T(g["logits"])
    rel = (logits.cpu() - ref).abs().max().item() / ref.abs().max().item()
    assert rel < 0.08 and abs(loss.item() - float(g["loss"])) < 3e-2
    a, b = eng.Gr["final_conv.weight"].cpu().flatten(), T(g["g_final_w"]).flatten()
    assert torch.dot(a, b) / (a.norm() * b.norm()) > 0.98
    # the nn.Module mirror with the reference's constructor arguments
    from mdeical_image_segmentation_amd.model.unet3d.model import UNet3D
    torch.manual_seed(0)
    m = UNet3D(1, 3, f_maps=[64, 128, 256], num_levels=3, upsample="deconv").cuda()
    lg = m(T(g["x"]).cuda())
    assert (lg.detach().cpu() - ref).abs().max().item() < 1e-4
    lg.sum().backward()
    assert m.decoders[0].upsampling.upsample.conv_transposed.weight.grad is not None
