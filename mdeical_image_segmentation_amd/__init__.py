"""MI355X-native U-Net segmentation hot path (HIP/CDNA4 kernels behind a C ABI) with the reference's
Python interface mirrored on top.  See DESIGN.md / INTEGRATION.md."""
from ._lib import MisError, LIB_PATH  # noqa: F401

__all__ = ["MisError", "LIB_PATH"]
