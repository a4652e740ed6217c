"""Mirror of dataset/unet2d_dataset/MYDataCollator.py:3-19: stack the samples' "image" / "mask" into the batch keys "images" / "labels" the model takes.
The samples already live in HBM (MYDataset mirror), so the stack is a device-to-device copy and the Trainer's host-to-device step has nothing to move."""
import torch


class DRIVEDataCollator:
    def __call__(self, batch):
        return {"images": torch.stack([s["image"] for s in batch], dim=0), "labels": torch.stack([s["mask"] for s in batch], dim=0)}


class BUSIDataCollator(DRIVEDataCollator):
    pass
