"""2-D sample pipeline (dataset/unet2d_dataset mirror): the fused device kernel against the numpy restatement of the albumentations chain
(oracle/dataset2d_oracle.py - parity unpinned, the library is absent), the file-based DRIVE / BUSI datasets (PNG decode once, HBM-resident samples,
the reference's split) and the collator."""
import itertools
import os

import numpy as np
import pytest
import torch

from oracle import dataset2d_oracle as do


def _sample(seed, H, W):
    r = np.random.RandomState(seed)
    return r.randint(0, 256, (H, W, 3)).astype(np.uint8), (r.rand(H, W) > 0.6).astype(np.uint8) * 255


def test_oracle_identity_and_geometry():
    img, mask = _sample(1, 6, 9)
    im, m = do.sample_pipeline(img, mask, size=(6, 9))
    assert im.shape == (3, 6, 9) and np.array_equal(im, img.transpose(2, 0, 1).astype(np.float32) / 255) and np.array_equal(m[0], mask / 255.0)
    im, _ = do.sample_pipeline(img, mask, size=(6, 9), hflip=True, vflip=True)        # both flips = a half turn
    im2, _ = do.sample_pipeline(img, mask, size=(6, 9), rot_k=2)
    assert np.array_equal(im, im2)
    im, _ = do.sample_pipeline(img, mask, size=(12, 18))                                 # exact 2x nearest up-sampling repeats pixels
    assert np.array_equal(im[:, ::2, ::2], img.transpose(2, 0, 1).astype(np.float32) / 255)
    im, _ = do.sample_pipeline(img, mask, size=(6, 9), bc=(1.2, 0.1))
    want = np.clip(np.arange(256, dtype=np.float32) * np.float32(1.2) + np.float32(25.5), 0, 255).astype(np.uint8)[img]
    assert np.array_equal(im, want.transpose(2, 0, 1).astype(np.float32) / 255)


@pytest.mark.gpu
def test_device_pipeline_matches_oracle_for_every_parameter_combination():
    from mdeical_image_segmentation_amd.dataset.unet2d_dataset import DeviceSegmentationDataset
    img, mask = _sample(2, 37, 53)
    for size in ((64, 64), (40, 72)):
        ds = DeviceSegmentationDataset([img], [mask], train=True, size=size)
        for hflip, vflip, rot_k, transpose, bc in itertools.product((False, True), (False, True), range(4), (False, True), (None, (0.83, 0.17), (1.19, -0.2))):
            p = {"hflip": hflip, "vflip": vflip, "rot_k": rot_k, "transpose": transpose, "bc": bc}
            got = ds.apply(0, p)
            wi, wm = do.sample_pipeline(img, mask, size=size, hflip=hflip, vflip=vflip, rot_k=rot_k, transpose=transpose, bc=bc)
            assert got["image"].shape == wi.shape and np.array_equal(got["image"].cpu().numpy(), wi), p
            assert np.array_equal(got["mask"].cpu().numpy(), wm), p


@pytest.mark.gpu
def test_file_datasets_split_collate_and_training_draws(tmp_path):
    from PIL import Image
    from sklearn.model_selection import train_test_split

    import mdeical_image_segmentation_amd.dropin as d
    d.install()
    from unet2d_dataset import BUSIDataCollator, BUSIDataset, DRIVEDataCollator, DRIVEDataset
    n = 10
    for sub in ("images", "labels", "mask/0"):
        os.makedirs(tmp_path / sub, exist_ok=True)
    samples = {}
    for i in range(n):
        img, mask = _sample(10 + i, 30 + i, 41)
        Image.fromarray(img).save(tmp_path / "images" / f"{i:02d}.png")
        Image.fromarray(mask).save(tmp_path / "labels" / f"{i:02d}.png")
        Image.fromarray(mask).save(tmp_path / "mask" / "0" / f"{i:02d}.png")
        samples[f"{i:02d}.png"] = (img, mask)
    names = sorted(samples)
    tr, tmp = train_test_split(names, test_size=(1 - 0.7), random_state=42)
    ev, te = train_test_split(tmp, test_size=(1 - 0.2 / (1 - 0.7)), random_state=42)
    for mode, want in (("train", tr), ("eval", ev), ("test", te)):
        ds = DRIVEDataset(str(tmp_path), mode=mode)
        assert [os.path.basename(p) for p in ds.images_path] == want and len(ds) == len(want)
        if mode != "train":                                   # deterministic: resize only
            for k, name in enumerate(want):
                s = ds[k]
                wi, wm = do.sample_pipeline(*samples[name])
                assert s["image"].is_cuda and s["image"].shape == (3, 512, 512) and s["mask"].shape == (1, 512, 512)
                assert np.array_equal(s["image"].cpu().numpy(), wi) and np.array_equal(s["mask"].cpu().numpy(), wm)
    ds = DRIVEDataset(str(tmp_path), mode="train", aug_seed=3)
    twin = DRIVEDataset(str(tmp_path), mode="train", aug_seed=3)
    batch = DRIVEDataCollator()([ds[k] for k in range(4)])
    assert batch["images"].shape == (4, 3, 512, 512) and batch["labels"].shape == (4, 1, 512, 512) and batch["images"].is_cuda
    for k in range(4):                                        # the same seed gives the same draws; every draw equals the oracle on its parameters
        p = twin.sample_params()
        wi, wm = do.sample_pipeline(*samples[os.path.basename(ds.images_path[k])], hflip=p["hflip"], vflip=p["vflip"], rot_k=p["rot_k"],
                                    transpose=p["transpose"], bc=p["bc"])
        assert np.array_equal(batch["images"][k].cpu().numpy(), wi) and np.array_equal(batch["labels"][k].cpu().numpy(), wm)
    b = BUSIDataset(str(tmp_path))
    assert len(b) == n and BUSIDataCollator()([b[0], b[1]])["images"].shape == (2, 3, 512, 512)
    with pytest.raises(NotImplementedError):
        DRIVEDataset(str(tmp_path), augmentations=object())
    with pytest.raises(ValueError):
        DRIVEDataset(str(tmp_path), mode="nope")
