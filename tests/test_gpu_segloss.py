"""SegmentationLoss (F1 + MS-SSIM + IoU; reference model/unet2d/loss.py) on the HIP kernels against the CPU oracle restatement: values of
all three terms, the gradient (torch autograd through the oracle), odd image sizes (padded pooling), and the UNetModel(UNet_3Plus) path."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("shape", [(2, 1, 176, 192), (1, 1, 203, 181)])
def test_segmentation_loss_value_and_gradient(shape):
    from mdeical_image_segmentation_amd.model.unet2d import loss as L
    from oracle import segloss_oracle as so
    gen = torch.Generator().manual_seed(shape[2])
    t = (torch.rand(*shape, generator=gen) > 0.6).float()
    x = (torch.randn(*shape, generator=gen) * 1.5 + (t * 2 - 1)).requires_grad_(True)
    ref = so.segmentation_loss(x, t)
    ref.backward()
    xd = x.detach().cuda().requires_grad_(True)
    td = t.cuda()
    loss = L.SegmentationLoss()(xd, td)
    assert abs(loss.item() - ref.item()) < 2e-5, (loss.item(), ref.item())
    loss.backward()
    g = x.grad
    assert (xd.grad.cpu() - g).abs().max().item() < 2e-3 * g.abs().max().item() + 1e-9, (xd.grad.cpu() - g).abs().max().item()
    rel = (xd.grad.cpu() - g).norm().item() / g.norm().item()
    assert rel < 1e-3, rel
    # the three terms on their own
    assert abs(L.F1Loss()(xd.detach(), td).item() - so.f1_loss(x.detach(), t).item()) < 1e-5
    assert abs(L.IoULoss()(xd.detach(), td).item() - so.iou_loss(x.detach(), t).item()) < 1e-5
    assert abs(L.MSSSIMLoss()(xd.detach(), td).item() - so.msssim_loss(x.detach(), t).item()) < 2e-5
    x2 = x.detach().clone().requires_grad_(True)
    so.msssim_loss(x2, t).backward()
    x3 = x.detach().cuda().requires_grad_(True)
    (L.MSSSIMLoss()(x3, td) * 2.0).backward()
    assert (x3.grad.cpu() - 2.0 * x2.grad).norm().item() / (2.0 * x2.grad).norm().item() < 1e-3


def test_unet_model_with_unet3plus_trains():
    import mdeical_image_segmentation_amd.dropin as d
    d.install()
    from unet2d import UNetConfig, UNetModel
    torch.manual_seed(0)
    m = UNetModel(UNetConfig(in_channels=3, out_channels=1, unet_type="UNet_3Plus")).cuda().train()
    gen = torch.Generator().manual_seed(1)
    images = torch.randn(1, 3, 176, 176, generator=gen).cuda()
    labels = (torch.rand(1, 1, 176, 176, generator=gen) > 0.5).float().cuda()
    out = m(images=images, labels=labels)
    assert out.logits.shape == (1, 1, 176, 176) and torch.isfinite(out.loss)
    out.loss.backward()
    gw = m.unet.outconv1.weight.grad
    assert gw is not None and torch.isfinite(gw).all() and gw.abs().sum() > 0
    with pytest.raises(Exception):
        from mdeical_image_segmentation_amd.model.unet2d.loss import SegmentationLoss
        SegmentationLoss()(torch.zeros(1, 1, 64, 64).cuda(), torch.zeros(1, 1, 64, 64).cuda())      # too small for 5 scales: loud


def test_f1_and_iou_loss_match_the_reference_classes():
    """the HIP F1Loss / IoULoss against fixture g19_segloss.npz = values and dL/dlogits of the REAL reference classes (model/unet2d/loss.py:32-56;
    tests/golden/make_golden_segloss.py), incl. images far smaller than the MS-SSIM term of SegmentationLoss accepts (the two terms alone are global sums)"""
    from conftest import load_golden
    from mdeical_image_segmentation_amd.model.unet2d import loss as L
    g = load_golden("g19_segloss.npz")
    for i in range(3):
        t = torch.from_numpy(g[f"t{i}"].astype(np.float32)).cuda()
        for name, cls in (("f1", L.F1Loss), ("iou", L.IoULoss)):
            x = torch.from_numpy(g[f"x{i}"]).cuda().requires_grad_(True)
            loss = cls()(x, t)
            loss.backward()
            assert abs(loss.item() - float(g[f"{name}{i}"])) < 2e-6, (name, i, loss.item(), float(g[f"{name}{i}"]))
            ref = torch.from_numpy(g[f"{name}{i}_grad"])
            err = (x.grad.cpu() - ref).abs().max().item()
            assert err <= 2e-5 * ref.abs().max().item() + 1e-12, (name, i, err, ref.abs().max().item())
