from . import utils
from .utils import FilterSliceBuilder, SliceBuilder, VolumeDataset, calculate_stats, get_slice_builder, get_train_loaders

__all__ = ["SliceBuilder", "FilterSliceBuilder", "VolumeDataset", "calculate_stats", "get_slice_builder", "get_train_loaders", "utils"]
