"""Mirror of dataset/unet3d_dataset/hdf5.py: file-backed patch datasets with the reference's constructor and `create_datasets` (:44-269, :271-340 Standard,
:342-400 Lazy).  The MI355X design keeps the whole volume in HBM (VolumeDataset): a file is read ONCE at construction, patches are device slices pushed through
the on-device Transformer, so `StandardHDF5Dataset` and `LazyHDF5Dataset` behave identically here (288 GB of HBM make the lazy variant's purpose moot).

File formats: `.h5 / .hdf5 / .hdf / .hd5` through h5py WHEN IT IS INSTALLED (it is not part of this image: a clear ImportError otherwise), and `.npz` archives
whose keys play the role of the HDF5 internal paths (`raw`, `label`, optional weight map) - the format the tests and INTEGRATION.md use."""
import glob
import os

import numpy as np

from ...model.unet3d.utils import get_logger
from .utils import VolumeDataset

logger = get_logger("HDF5Dataset")

_H5_EXT = (".h5", ".hdf", ".hdf5", ".hd5")
_EXTS = _H5_EXT + (".npz",)


def traverse_h5_paths(file_paths):
    """hdf5.py:27-41: directories are expanded to the volume files inside them"""
    assert isinstance(file_paths, list)
    results = []
    for file_path in file_paths:
        if os.path.isdir(file_path):
            for ext in _EXTS:
                results.extend(sorted(glob.glob(os.path.join(file_path, "*" + ext))))
        else:
            results.append(file_path)
    return results


def read_volumes(file_path, internal_paths):
    """{internal path: ndarray} of one file; None paths are skipped"""
    wanted = [p for p in internal_paths if p is not None]
    if str(file_path).lower().endswith(".npz"):
        with np.load(file_path) as f:
            return {p: np.asarray(f[p]) for p in wanted}
    if str(file_path).lower().endswith(_H5_EXT):
        try:
            import h5py
        except ImportError as e:
            raise ImportError(f"reading {file_path} needs h5py, which is not installed here: convert the volumes to an .npz archive "
                              "(np.savez(path, raw=..., label=...)) or install h5py") from e
        with h5py.File(file_path, "r") as f:
            return {p: f[p][:] for p in wanted}
    raise ValueError(f"unsupported volume file {file_path}: expected one of {_EXTS}")


class AbstractHDF5Dataset(VolumeDataset):
    def __init__(self, file_path, phase, slice_builder_config, transformer_config, raw_internal_path="raw", label_internal_path="label",
                 weight_internal_path=None, global_normalization=True, device="cuda"):
        self.file_path = file_path
        self.raw_internal_path, self.label_internal_path, self.weight_internal_path = raw_internal_path, label_internal_path, weight_internal_path
        vols = read_volumes(file_path, [raw_internal_path, label_internal_path if phase != "test" else None, weight_internal_path if phase != "test" else None])
        super().__init__(vols[raw_internal_path], vols.get(label_internal_path) if phase != "test" else None, phase, slice_builder_config,
                         transformer_config, weight_map=vols.get(weight_internal_path) if weight_internal_path is not None else None,
                         global_normalization=bool(global_normalization), device=device)
        logger.info(f"Number of patches: {self.patch_count}")

    @classmethod
    def create_datasets(cls, dataset_config, phase):
        """hdf5.py:231-268: one dataset per file of `dataset_config[phase]['file_paths']`; a file that fails to load is skipped with an error log"""
        phase_config = dataset_config[phase]
        transformer_config, slice_builder_config = phase_config["transformer"], phase_config["slice_builder"]
        datasets = []
        for file_path in traverse_h5_paths(phase_config["file_paths"]):
            try:
                logger.info(f"Loading {phase} set from: {file_path}...")
                datasets.append(cls(file_path=file_path, phase=phase, slice_builder_config=slice_builder_config, transformer_config=transformer_config,
                                    raw_internal_path=dataset_config.get("raw_internal_path", "raw"),
                                    label_internal_path=dataset_config.get("label_internal_path", "label"),
                                    weight_internal_path=dataset_config.get("weight_internal_path", None),
                                    global_normalization=dataset_config.get("global_normalization", None)))
            except Exception:
                logger.error(f"Skipping {phase} set: {file_path}", exc_info=True)
        return datasets


class StandardHDF5Dataset(AbstractHDF5Dataset):
    pass


class LazyHDF5Dataset(AbstractHDF5Dataset):
    pass
