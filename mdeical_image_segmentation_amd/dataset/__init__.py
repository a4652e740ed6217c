"""Mirror of the reference's `dataset` package for the 2-D path: HBM-resident datasets whose per-sample pipeline runs on the MI355X."""
from . import unet2d_dataset, unet3d_dataset
from .unet2d_dataset import *  # noqa: F401,F403
from .unet3d_dataset import *  # noqa: F401,F403
