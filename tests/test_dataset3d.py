"""3-D dataset side: the slice builders against the positions the REAL reference classes produce (tests/golden/g14_slices.npz), calculate_stats, and
the HBM-resident VolumeDataset (patches cut on the device and pushed through the on-device Transformer)."""
import numpy as np
import pytest
import torch

from conftest import load_golden


def starts(slices):
    return np.array([[s.start for s in idx] + [s.stop for s in idx] for idx in slices], dtype=np.int64)


def test_slice_builders_match_reference_positions():
    from mdeical_image_segmentation_amd.dataset.unet3d_dataset import FilterSliceBuilder, SliceBuilder, calculate_stats, get_slice_builder
    g = load_golden("g14_slices.npz")
    raw3, raw4, lab3 = np.zeros((30, 100, 90), np.float32), np.zeros((2, 17, 70, 131), np.float32), g["lab3"].astype(np.int64)
    sb = SliceBuilder(raw3, lab3, None, (8, 64, 64), (4, 32, 40))
    assert np.array_equal(starts(sb.raw_slices), g["sb3_raw"]) and np.array_equal(starts(sb.label_slices), g["sb3_label"]) and sb.weight_slices is None
    assert np.array_equal(starts(SliceBuilder(raw4, None, None, (17, 64, 64), (17, 64, 64)).raw_slices), g["sb4_raw"])
    assert np.array_equal(starts(SliceBuilder(raw3, None, None, (5, 7, 9), (5, 6, 4), skip_shape_check=True).raw_slices), g["sb_small"])
    for tag, kw in (("a", dict(threshold=0.3, slack_acceptance=0.2)), ("b", dict(ignore_index=-1, threshold=0.25, slack_acceptance=0.05))):
        fb = FilterSliceBuilder(raw3, lab3, None, (8, 64, 64), (4, 32, 40), **kw)
        assert np.array_equal(starts(fb.raw_slices), g[f"fb_{tag}_raw"]) and np.array_equal(starts(fb.label_slices), g[f"fb_{tag}_label"]), tag
    fb = get_slice_builder(raw3, lab3, None, {"name": "FilterSliceBuilder", "patch_shape": (8, 64, 64), "stride_shape": (4, 32, 40), "threshold": 0.3,
                                              "slack_acceptance": 0.2})
    assert np.array_equal(starts(fb.raw_slices), g["fb_a_raw"])
    with pytest.raises(AssertionError):
        SliceBuilder(raw3, None, None, (8, 32, 64), (4, 32, 40))           # height / width >= 64 unless skip_shape_check
    with pytest.raises(AssertionError):
        SliceBuilder(raw3, None, None, (31, 64, 64), (4, 32, 40))          # the patch does not fit
    st = calculate_stats(g["stats_in"])
    assert np.allclose([st["pmin"], st["pmax"], st["mean"], st["std"]], g["stats"], rtol=1e-12, atol=0)
    assert calculate_stats(None, True) == {"pmin": None, "pmax": None, "mean": None, "std": None}


@pytest.mark.gpu
def test_volume_dataset_patches_and_transforms_on_the_device():
    from mdeical_image_segmentation_amd.augment.unet3d_augment import transforms as tr
    from mdeical_image_segmentation_amd.dataset.unet3d_dataset import VolumeDataset
    rng = np.random.RandomState(3)
    raw = rng.randn(20, 80, 72).astype(np.float32)
    lab = (rng.rand(20, 80, 72) * 3).astype(np.int64)
    sbc = {"name": "SliceBuilder", "patch_shape": (8, 64, 64), "stride_shape": (8, 16, 8)}
    tcfg = {"raw": [{"name": "Standardize"}, {"name": "RandomFlip"}, {"name": "RandomRotate90"}, {"name": "ToTensor", "expand_dims": True}],
            "label": [{"name": "RandomFlip"}, {"name": "RandomRotate90"}, {"name": "ToTensor", "expand_dims": False, "dtype": "int64"}]}
    tr.GLOBAL_RANDOM_STATE = np.random.RandomState(47)
    ds = VolumeDataset(raw, lab, "train", sbc, tcfg)
    assert len(ds) == len(ds.raw_slices) == 3 * 2 * 2 and ds.volume_shape() == (20, 80, 72)
    # the same Transformer seed on the host: replay every sample with the numpy oracle of the transforms
    from oracle import augment_oracle as ao
    seed = ds.transformer.seed
    rs_raw, rs_lab = (np.random.RandomState(seed), np.random.RandomState(seed)), (np.random.RandomState(seed), np.random.RandomState(seed))
    mean, std = raw.mean(), raw.std()

    def flip_rot(m, rs_f, rs_r):
        mask = sum(1 << ax for ax in range(3) if rs_f.uniform() > 0.5)
        m = ao.flip(m, mask)
        return ao.rot90(m, rs_r.randint(0, 4))

    for i in range(len(ds)):
        r, l = ds[i]
        assert r.is_cuda and l.is_cuda and tuple(r.shape) == (1, 8, 64, 64) and tuple(l.shape) == (8, 64, 64)
        want_r = flip_rot(ao.standardize(raw[ds.raw_slices[i]], mean, std), *rs_raw)
        want_l = flip_rot(lab[ds.label_slices[i]], *rs_lab)
        assert np.allclose(r[0].cpu().numpy(), want_r, atol=2e-6), i
        assert np.array_equal(l.cpu().numpy(), want_l), i
    test = VolumeDataset(raw, None, "test", dict(sbc, halo_shape=[2, 4, 4]), {"raw": [{"name": "ToTensor", "expand_dims": True}]}, global_normalization=False)
    p, idx = test[5]
    assert tuple(p.shape) == (1, 12, 72, 72) and len(idx) == 3
    padded = np.pad(raw, [(2, 2), (4, 4), (4, 4)], mode="reflect")
    z, y, x = idx
    assert np.array_equal(p[0].cpu().numpy(), padded[z.start:z.stop + 4, y.start:y.stop + 8, x.start:x.stop + 8])
