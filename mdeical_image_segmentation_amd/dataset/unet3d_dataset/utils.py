"""Mirror of the patch side of the reference's 3-D datasets: `SliceBuilder` / `FilterSliceBuilder` (dataset/unet3d_dataset/utils.py:47-157),
`calculate_stats` (:290-311), `get_slice_builder` (:168-172), and `VolumeDataset`, the HBM-resident counterpart of `AbstractHDF5Dataset`
(dataset/unet3d_dataset/hdf5.py:44-200): the raw / label volumes live in device memory, a sample is a patch cut from them and pushed through the
on-device `Transformer` (augment mirror) - nothing is read from disk or moved over PCIe per sample.  File-backed datasets (`hdf5.py`) read the
volumes once and hand them to `VolumeDataset`; `get_train_loaders` (:182-227) builds the loaders `create_trainer` needs.  The slice bookkeeping is one-time host logic, identical to the reference's (golden `tests/golden/g14_slices.npz`)."""
import numpy as np
import torch

from ...augment.unet3d_augment import transforms
from ..._lib import MisError


class SliceBuilder:
    """utils.py:47-128: all patch positions of a (C,)D,H,W volume for a patch / stride shape; a last patch flush with the border is added per axis"""

    def __init__(self, raw_dataset, label_dataset, weight_dataset, patch_shape, stride_shape, **kwargs):
        patch_shape, stride_shape = tuple(patch_shape), tuple(stride_shape)
        skip_shape_check = kwargs.get("skip_shape_check", False)
        if not skip_shape_check:
            self._check_patch_shape(patch_shape)
        self._raw_slices = self._build_slices(raw_dataset, patch_shape, stride_shape)
        self._label_slices = None if label_dataset is None else self._build_slices(label_dataset, patch_shape, stride_shape)
        if self._label_slices is not None:
            assert len(self._raw_slices) == len(self._label_slices)
        self._weight_slices = None if weight_dataset is None else self._build_slices(weight_dataset, patch_shape, stride_shape)
        if self._weight_slices is not None:
            assert len(self._raw_slices) == len(self._weight_slices)

    raw_slices = property(lambda self: self._raw_slices)
    label_slices = property(lambda self: self._label_slices)
    weight_slices = property(lambda self: self._weight_slices)

    @staticmethod
    def _build_slices(dataset, patch_shape, stride_shape):
        shape = tuple(dataset.shape)
        channels = shape[0] if len(shape) == 4 else None
        spatial = shape[-3:]
        out = []
        for z in SliceBuilder._gen_indices(spatial[0], patch_shape[0], stride_shape[0]):
            for y in SliceBuilder._gen_indices(spatial[1], patch_shape[1], stride_shape[1]):
                for x in SliceBuilder._gen_indices(spatial[2], patch_shape[2], stride_shape[2]):
                    idx = (slice(z, z + patch_shape[0]), slice(y, y + patch_shape[1]), slice(x, x + patch_shape[2]))
                    out.append(idx if channels is None else (slice(0, channels),) + idx)
        return out

    @staticmethod
    def _gen_indices(i, k, s):
        assert i >= k, "Sample size has to be bigger than the patch size"
        j = 0
        for j in range(0, i - k + 1, s):
            yield j
        if j + k < i:
            yield i - k

    @staticmethod
    def _check_patch_shape(patch_shape):
        assert len(patch_shape) == 3, "patch_shape must be a 3D tuple"
        assert patch_shape[1] >= 64 and patch_shape[2] >= 64, "Height and Width must be greater or equal 64"


class FilterSliceBuilder(SliceBuilder):
    """utils.py:131-157: drops patches whose label content (non-zero, non-ignored voxels) is <= threshold, except with probability slack_acceptance
    (RandomState(47), one draw per rejected candidate exactly as the reference's short-circuit `or` makes them)"""

    def __init__(self, raw_dataset, label_dataset, weight_dataset, patch_shape, stride_shape, ignore_index=None, threshold=0.6,
                 slack_acceptance=0.01, **kwargs):
        super().__init__(raw_dataset, label_dataset, weight_dataset, patch_shape, stride_shape, **kwargs)
        if label_dataset is None:
            return
        rand_state = np.random.RandomState(47)
        lab = label_dataset.detach().cpu().numpy() if isinstance(label_dataset, torch.Tensor) else np.asarray(label_dataset)

        def keep(label_idx):
            patch = lab[label_idx]
            if ignore_index is not None:
                patch = np.where(patch == ignore_index, 0, patch)
            return np.count_nonzero(patch != 0) / patch.size > threshold or rand_state.rand() < slack_acceptance

        kept = [(r, l) for r, l in zip(self.raw_slices, self.label_slices) if keep(l)]
        self._raw_slices = [r for r, _ in kept]
        self._label_slices = [l for _, l in kept]


def get_slice_builder(raws, labels, weight_maps, config):
    assert "name" in config
    classes = {"SliceBuilder": SliceBuilder, "FilterSliceBuilder": FilterSliceBuilder}
    if config["name"] not in classes:
        raise NotImplementedError(f"slice builder {config['name']}")
    return classes[config["name"]](raws, labels, weight_maps, **config)


def calculate_stats(img, skip=False):
    """utils.py:290-311 (one-time host pass over the raw volume)"""
    if skip:
        return {"pmin": None, "pmax": None, "mean": None, "std": None}
    img = img.detach().cpu().numpy() if isinstance(img, torch.Tensor) else np.asarray(img)
    return {"pmin": np.percentile(img, 1), "pmax": np.percentile(img, 99.6), "mean": np.mean(img), "std": np.std(img)}


class VolumeDataset(torch.utils.data.Dataset):
    """hdf5.py:44-200 with the volumes in HBM instead of an HDF5 file: raw (D,H,W) or (C,D,H,W) float, label (optional) of the same spatial size.
    phase 'train' / 'val': (raw_transform(raw patch), label_transform(label patch)[, weight]); 'test': (raw_transform(padded raw patch), raw index)."""

    def __init__(self, raw, label, phase, slice_builder_config, transformer_config, weight_map=None, global_normalization=True, device="cuda"):
        assert phase in ["train", "val", "test"]
        self.phase = phase
        self.device = torch.device(device)
        self.halo_shape = slice_builder_config.get("halo_shape", [0, 0, 0])
        stats = calculate_stats(raw, skip=not global_normalization)
        self.transformer = transforms.Transformer(transformer_config, stats)
        self.raw_transform = self.transformer.raw_transform()
        self.raw = self._up(raw)
        if phase != "test":
            if label is None:
                raise MisError("VolumeDataset: a label volume is required in the 'train' and 'val' phases")
            self.label_transform = self.transformer.label_transform()
            self.weight_transform = self.transformer.weight_transform() if weight_map is not None else None
            self.label = self._up(label)
            self.weight_map = None if weight_map is None else self._up(weight_map)
            assert self.raw.dim() in [3, 4], "Raw dataset must be 3D (DxHxW) or 4D (CxDxHxW)"
            assert self.label.dim() in [3, 4], "Label dataset must be 3D (DxHxW) or 4D (CxDxHxW)"
            assert tuple(self.raw.shape[-3:]) == tuple(self.label.shape[-3:]), "Raw and labels have to be of the same size"
        else:
            self.label = self.weight_map = None
            if sum(self.halo_shape) != 0:           # 'test' patches carry their halo: mirror-pad the volume once (utils.py:314-361)
                pad = [(0, 0)] * (self.raw.dim() - 3) + [(h, h) for h in self.halo_shape]
                self.raw_padded = self._up(np.pad(self.raw.cpu().numpy(), pad, mode="reflect"))
            else:
                self.raw_padded = self.raw
        sb = get_slice_builder(self.raw, self.label, self.weight_map, slice_builder_config)
        self.raw_slices, self.label_slices, self.weight_slices = sb.raw_slices, sb.label_slices, sb.weight_slices
        self.patch_count = len(self.raw_slices)

    def _up(self, a):
        t = a if isinstance(a, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(a))
        return t.to(self.device)

    def volume_shape(self):
        return tuple(self.raw.shape[-3:])

    def __getitem__(self, idx):
        if idx >= len(self):
            raise StopIteration
        raw_idx = self.raw_slices[idx]
        if self.phase == "test":
            spatial = raw_idx[-3:]
            padded = tuple(slice(i.start, i.stop + 2 * h) for i, h in zip(spatial, self.halo_shape))
            if len(raw_idx) == 4:
                padded = (slice(None),) + padded
            return self.raw_transform(self.raw_padded[padded].contiguous()), spatial
        raw = self.raw_transform(self.raw[raw_idx].contiguous())
        label = self.label_transform(self.label[self.label_slices[idx]].contiguous())
        if self.weight_map is not None:
            return raw, label, self.weight_transform(self.weight_map[self.weight_slices[idx]].contiguous())
        return raw, label

    def __len__(self):
        return self.patch_count


def _loader_classes(class_name):
    """utils.py:166-172"""
    from . import hdf5
    if not hasattr(hdf5, class_name):
        raise RuntimeError(f"Unsupported dataset class: {class_name}")
    return getattr(hdf5, class_name)


def get_train_loaders(config):
    """utils.py:182-227: {'train': DataLoader, 'val': DataLoader} over ConcatDataset(create_datasets(...)).  The samples are device tensors (the volumes live in
    HBM), so the loaders run in the training process (num_workers and pin_memory of the config do not apply); batch_size is the config's, per process - data
    parallelism is one process per GPU here, not nn.DataParallel's batch_size * device_count."""
    from torch.utils.data import ConcatDataset, DataLoader
    assert "loaders" in config, "Could not find data loaders configuration"
    loaders_config = config["loaders"]
    dataset_class = _loader_classes(loaders_config.get("dataset", None) or "StandardHDF5Dataset")
    assert set(loaders_config["train"]["file_paths"]).isdisjoint(loaders_config["val"]["file_paths"]), \
        "Train and validation 'file_paths' overlap. One cannot use validation data for training!"
    train_datasets = dataset_class.create_datasets(loaders_config, phase="train")
    val_datasets = dataset_class.create_datasets(loaders_config, phase="val")
    batch_size = loaders_config.get("batch_size", 1)
    return {"train": DataLoader(ConcatDataset(train_datasets), batch_size=batch_size, shuffle=True, num_workers=0),
            "val": DataLoader(ConcatDataset(val_datasets), batch_size=batch_size, shuffle=False, num_workers=0)}
