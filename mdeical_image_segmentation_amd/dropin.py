"""Make the reference's import surface resolve to this package:

    import mdeical_image_segmentation_amd.dropin as d; d.install()
    from unet2d import UNetModel, UNetConfig          # train.py:5
    from trainer import CustomTrainer, compute_metrics # train.py:9
    from model import UNetModel                        # test_trainer.py:6
    from unet2d_dataset import DRIVEDataset, DRIVEDataCollator   # train.py:6 (HBM-resident samples, device pipeline)
"""
import importlib
import sys


def install():
    pkg = __name__.rsplit(".", 1)[0]
    for alias, target in (("model", f"{pkg}.model"), ("model.unet2d", f"{pkg}.model.unet2d"),
                          ("model.unet3d", f"{pkg}.model.unet3d"), ("unet2d", f"{pkg}.model.unet2d"),
                          ("unet3d", f"{pkg}.model.unet3d"), ("trainer", f"{pkg}.trainer"), ("dataset", f"{pkg}.dataset"),
                          ("unet2d_dataset", f"{pkg}.dataset.unet2d_dataset")):
        sys.modules[alias] = importlib.import_module(target)
