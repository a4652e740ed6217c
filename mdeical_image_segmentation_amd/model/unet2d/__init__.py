"""Mirror of the reference's `model.unet2d` package (model/unet2d/__init__.py:1-4)."""
from . import init_weights as _iw, layers, unet  # noqa: F401
from .init_weights import init_weights  # noqa: F401
from .layers import *  # noqa: F401,F403
from .unet import (UNet, UNet_3Plus, UNet_3Plus_DeepSup, UNet_3Plus_DeepSup_CGM, UNetConfig, UNetModel,  # noqa: F401
                   UNetModelOutput)
