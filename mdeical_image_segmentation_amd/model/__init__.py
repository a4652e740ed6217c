"""Mirror of the reference's `model` package (model/__init__.py:1-4)."""
from . import unet2d, unet3d  # noqa: F401
from .unet2d import *  # noqa: F401,F403
from .unet3d import *  # noqa: F401,F403
