"""Mirror of the reference's 2-D datasets (dataset/unet2d_dataset/MYDataset.py:52-176: `DRIVEDataset`, `BUSIDataset`) with the decoded samples
RESIDENT IN HBM and the per-sample pipeline on the device (csrc/dataset2d.hip, one fused gather per sample):

    train: Resize(512, 512, nearest) -> HorizontalFlip(.5) -> VerticalFlip(.5) -> RandomRotate90(.5) -> Transpose(.5) -> RandomBrightnessContrast(.5)
           -> CHW float / 255           (MYDataset.py:127-141)           eval / test: Resize only (:100-115)

Same constructor arguments, file layout (`images/*`, `labels/*`; BUSI: `images/*`, `mask/0/*`), sorted listing and sklearn train/eval/test split
(:73-96), same sample dict {"image": (3, 512, 512) float, "mask": (1, 512, 512) float}.  Differences, all documented:
  * files are decoded ONCE at construction (PIL, host) and uploaded as uint8; `__getitem__` returns CUDA tensors - keep `dataloader_num_workers=0`
    and `dataloader_pin_memory=False`; with the collator mirror a batch never touches the host (738 vs 94 img/s under the HF Trainer);
  * albumentations (1.4.10 in the reference's requirements) is a third-party library that is absent here: its published algorithm is restated
    (oracle/dataset2d_oracle.py, "parity unpinned") and the random parameters come from a numpy RandomState of this class (`aug_seed`), not
    from albumentations' use of Python's global `random` module - same distributions, different draws;
  * a custom `augmentations` object (an albumentations.Compose in the reference) cannot run on the device: only None (the default chain) is accepted."""
import os
from glob import glob

import numpy as np
import torch
from torch.utils.data import Dataset

from ..._lib import MisError, check, load, stream_ptr

_SIZE = 512


def _decode(path, mode):
    from PIL import Image
    return np.ascontiguousarray(np.array(Image.open(path).convert(mode)))


class DeviceSegmentationDataset(Dataset):
    """uint8 samples in HBM + the device pipeline.  images: sequence of (H, W, 3) uint8 arrays / tensors, masks: sequence of (H, W) uint8."""

    def __init__(self, images, masks, train=True, aug_seed=0, size=(_SIZE, _SIZE), device="cuda"):
        if len(images) != len(masks):
            raise ValueError("The number of images and masks do not match.")
        self.device = torch.device(device)
        self.images = [self._up(a, 3) for a in images]
        self.masks = [self._up(a, 2) for a in masks]
        for im, mk in zip(self.images, self.masks):
            if tuple(im.shape[:2]) != tuple(mk.shape):
                raise MisError(f"image {tuple(im.shape)} and mask {tuple(mk.shape)} sizes differ")
        self.train = train
        self.size = size
        self.rng = np.random.RandomState(aug_seed)
        self.n_samples = len(self.images)

    def _up(self, a, ndim):
        t = torch.as_tensor(np.asarray(a) if not isinstance(a, torch.Tensor) else a)
        if t.dtype != torch.uint8 or t.dim() != ndim:
            raise MisError(f"expected a uint8 array with {ndim} dims, got {t.dtype} {tuple(t.shape)}")
        return t.contiguous().to(self.device)

    def sample_params(self):
        """one draw of the training chain's parameters (albumentations 1.4.10 distributions): flips / rot90 / transpose with p = 0.5, quarter turns
        uniform in {0..3}, brightness/contrast with p = 0.5, alpha = 1 + U(-0.2, 0.2), beta = U(-0.2, 0.2)"""
        r = self.rng
        p = {"hflip": r.rand() < 0.5, "vflip": r.rand() < 0.5, "rot_k": 0, "transpose": False, "bc": None}
        if r.rand() < 0.5:
            p["rot_k"] = int(r.randint(0, 4))
        p["transpose"] = r.rand() < 0.5
        if r.rand() < 0.5:
            p["bc"] = (1.0 + r.uniform(-0.2, 0.2), r.uniform(-0.2, 0.2))
        return p

    def apply(self, index, params):
        img, mask = self.images[index], self.masks[index]
        H, W, C = img.shape
        OH, OW = self.size
        swap = (params["rot_k"] & 1) != int(bool(params["transpose"]))
        FH, FW = (OW, OH) if swap else (OH, OW)
        out_i = torch.empty(C, FH, FW, dtype=torch.float32, device=self.device)
        out_m = torch.empty(1, FH, FW, dtype=torch.float32, device=self.device)
        bc = params["bc"]
        check(load().mis_aug2d_u8(img.data_ptr(), mask.data_ptr(), H, W, C, OH, OW, int(params["hflip"]), int(params["vflip"]), int(params["rot_k"]),
                                  int(params["transpose"]), 0 if bc is None else 1, 1.0 if bc is None else float(bc[0]), 0.0 if bc is None else float(bc[1]),
                                  out_i.data_ptr(), out_m.data_ptr(), stream_ptr()), "mis_aug2d_u8")
        return {"image": out_i, "mask": out_m}

    def __getitem__(self, index):
        params = self.sample_params() if self.train else {"hflip": False, "vflip": False, "rot_k": 0, "transpose": False, "bc": None}
        return self.apply(index, params)

    def __len__(self):
        return self.n_samples


class DRIVEDataset(DeviceSegmentationDataset):
    def __init__(self, data_path, augmentations=None, mode="train", train_ratio=0.7, eval_ratio=0.2, random_seed=42, aug_seed=0, device="cuda"):
        from sklearn.model_selection import train_test_split
        if augmentations is not None:
            raise NotImplementedError("a custom albumentations pipeline cannot run on the device: pass augmentations=None (the reference's default chain)")
        images_path = sorted(glob(os.path.join(data_path, "images", "*")))
        masks_path = sorted(glob(os.path.join(data_path, "labels", "*")))
        if not len(images_path) == len(masks_path):
            raise ValueError("The number of images and masks do not match.")
        train_images, temp_images, train_masks, temp_masks = train_test_split(images_path, masks_path, test_size=(1 - train_ratio), random_state=random_seed)
        eval_size = eval_ratio / (1 - train_ratio)
        eval_images, test_images, eval_masks, test_masks = train_test_split(temp_images, temp_masks, test_size=(1 - eval_size), random_state=random_seed)
        if mode == "train":
            self.images_path, self.masks_path = train_images, train_masks
        elif mode == "eval":
            self.images_path, self.masks_path = eval_images, eval_masks
        elif mode == "test":
            self.images_path, self.masks_path = test_images, test_masks
        else:
            raise ValueError("Mode should be 'train', 'eval', or 'test'.")
        super().__init__([_decode(p, "RGB") for p in self.images_path], [_decode(p, "L") for p in self.masks_path], train=(mode == "train"),
                         aug_seed=aug_seed, device=device)


class BUSIDataset(DeviceSegmentationDataset):
    def __init__(self, data_path, augmentations=None, aug_seed=0, device="cuda"):
        if augmentations is not None:
            raise NotImplementedError("a custom albumentations pipeline cannot run on the device: pass augmentations=None (the reference's default chain)")
        self.images_path = sorted(glob(os.path.join(data_path, "images", "*")))
        self.masks_path = sorted(glob(os.path.join(data_path, "mask", "0", "*")))
        super().__init__([_decode(p, "RGB") for p in self.images_path], [_decode(p, "L") for p in self.masks_path], train=True, aug_seed=aug_seed,
                         device=device)
