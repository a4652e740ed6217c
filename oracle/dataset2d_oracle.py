"""CPU restatement (numpy) of the reference's 2-D sample pipeline, dataset/unet2d_dataset/MYDataset.py:127-157 - test infrastructure only.

PARITY UNPINNED: the pipeline is albumentations 1.4.10 + opencv 4.9 (requirements.txt:7,95), third-party packages that are absent from this image and
from /root/reference, so no golden vector can be generated; the published algorithms are restated: cv2.resize INTER_NEAREST (source index
min(floor(dst * (1 / (dsize / ssize))), ssize - 1)), HorizontalFlip / VerticalFlip = cv2.flip, RandomRotate90 = np.rot90(img, factor), Transpose =
axis swap, RandomBrightnessContrast on uint8 = look-up table clip(arange(256, float32) * alpha + beta * 255, 0, 255).astype(uint8) (brightness_by_max),
ToTensorV2 = HWC -> CHW, then `.float() / 255` (:150-151)."""
import numpy as np


def resize_nearest(a, OH, OW):
    H, W = a.shape[:2]
    iy = np.minimum(np.floor(np.arange(OH) * (1.0 / (OH / H))).astype(np.int64), H - 1)
    ix = np.minimum(np.floor(np.arange(OW) * (1.0 / (OW / W))).astype(np.int64), W - 1)
    return a[iy[:, None], ix[None, :]]


def sample_pipeline(img, mask, size=(512, 512), hflip=False, vflip=False, rot_k=0, transpose=False, bc=None):
    """img uint8 (H, W, 3), mask uint8 (H, W) -> image float32 (3, FH, FW), mask float32 (1, FH, FW)"""
    out = []
    for a, is_img in ((img, True), (mask, False)):
        a = resize_nearest(a, *size)
        if hflip:
            a = a[:, ::-1]
        if vflip:
            a = a[::-1]
        a = np.rot90(a, rot_k)
        if transpose:
            a = a.transpose(1, 0, 2) if a.ndim == 3 else a.transpose(1, 0)
        if is_img and bc is not None:
            lut = np.arange(0, 256).astype("float32")
            lut *= np.float32(bc[0])
            lut += np.float32(bc[1] * 255)
            a = np.clip(lut, 0, 255).astype(np.uint8)[a]
        out.append(np.ascontiguousarray(a))
    image = out[0].transpose(2, 0, 1).astype(np.float32) / np.float32(255)
    m = out[1].astype(np.float32)[None] / np.float32(255)
    return image, m
