"""3-D validation metrics on the HIP path (`model/unet3d/metrics.py` mirror: MeanIoU, DiceCoefficient) against golden values from the real
reference classes (tests/golden/g11_metrics3d.npz): one-hot and label targets, ties, skip_channels, ignore_index (incl. the in-place zeroing of
the caller's target), single-channel thresholding."""
import numpy as np
import pytest
import torch

from conftest import load_golden

pytestmark = pytest.mark.gpu


def T(a):
    return torch.from_numpy(np.asarray(a)).cuda()


def test_mean_iou_and_dice_match_reference_goldens():
    from mdeical_image_segmentation_amd.model.unet3d.metrics import DiceCoefficient, MeanIoU, get_evaluation_metric
    g = load_golden("g11_metrics3d.npz")
    probs = T(g["probs"])

    def same(v, key):
        assert isinstance(v, torch.Tensor) and v.dtype == torch.float32 and v.dim() == 0
        assert v.item() == float(g[key]), (key, v.item(), float(g[key]))

    same(MeanIoU()(probs, T(g["onehot"])), "miou_onehot")
    same(MeanIoU()(probs, T(g["labels"])), "miou_labels")
    same(MeanIoU(skip_channels=(0,))(probs, T(g["labels"])), "miou_skip0")
    same(MeanIoU(ignore_index=-1)(probs, T(g["lab_ign"])), "miou_lab_ign")
    tmut = T(g["oh_ign"].copy())
    same(MeanIoU(ignore_index=-1)(probs, tmut), "miou_oh_ign")
    assert np.array_equal(tmut.cpu().numpy(), g["oh_ign_after"])
    same(MeanIoU()(T(g["p1"]), T(g["t1"])), "miou_c1")
    same(get_evaluation_metric({"eval_metric": {"name": "MeanIoU", "skip_channels": (0,)}})(probs, T(g["labels"])), "miou_skip0")
    d = DiceCoefficient()(probs, T(g["onehot"]))
    assert abs(d.item() - float(g["dice"])) < 2e-6
    assert abs(DiceCoefficient()(T(g["p1"]), T(g["t1"])).item() - float(g["dice_c1"])) < 2e-6
    with pytest.raises(Exception):
        MeanIoU()(probs.cpu(), T(g["labels"]))


def test_mean_iou_counts_at_volume_size():
    """128^3 x 3 channels: the integer counts equal torch's on the device (size-independent exactness of the atomics / first-maximum rule)"""
    from mdeical_image_segmentation_amd.model.unet3d.metrics import MeanIoU
    gen = torch.Generator(device="cuda").manual_seed(5)
    p = torch.rand(2, 3, 128, 128, 128, device="cuda", generator=gen)
    p = (p * 16).floor() / 16                                    # many exact ties
    lab = torch.randint(0, 3, (2, 128, 128, 128), device="cuda", generator=gen)
    got = MeanIoU()(p, lab).item()
    pred = torch.zeros_like(p, dtype=torch.uint8).scatter_(1, p.max(dim=1, keepdim=True)[1], 1)
    tgt = torch.zeros_like(p, dtype=torch.uint8).scatter_(1, lab.unsqueeze(1), 1)
    inter = (pred & tgt).flatten(2).sum(2).float()
    uni = (pred | tgt).flatten(2).sum(2).float().clamp(min=1e-8)
    want = (inter / uni).mean(1).mean().item()
    assert abs(got - want) < 1e-7, (got, want)
