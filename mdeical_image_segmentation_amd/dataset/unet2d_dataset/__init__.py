from . import MYDataCollator, MYDataset
from .MYDataCollator import BUSIDataCollator, DRIVEDataCollator
from .MYDataset import BUSIDataset, DeviceSegmentationDataset, DRIVEDataset

__all__ = ["DRIVEDataset", "BUSIDataset", "DeviceSegmentationDataset", "DRIVEDataCollator", "BUSIDataCollator", "MYDataset", "MYDataCollator"]
