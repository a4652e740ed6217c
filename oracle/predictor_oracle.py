"""CPU restatement (numpy) of the reference's patch-tiled prediction, model/unet3d/predictor.py:85-168 with
dataset/unet3d_dataset/utils.py:85-125 (SliceBuilder), :314-342 (mirror_pad), :345-361 (remove_padding).  TEST INFRASTRUCTURE ONLY.
Pinned by tests/golden/g8_predictor.npz (the reference's own SliceBuilder / mirror_pad / remove_padding / UNet3D)."""
import numpy as np


def gen_indices(i, k, s):
    """utils.py:110-116"""
    assert i >= k
    j = 0
    for j in range(0, i - k + 1, s):
        yield j
    if j + k < i:
        yield i - k


def build_slices(shape, patch_shape, stride_shape):
    """utils.py:85-108 for a 3-D dataset"""
    (iz, iy, ix), (kz, ky, kx), (sz, sy, sx) = shape, patch_shape, stride_shape
    return [(slice(z, z + kz), slice(y, y + ky), slice(x, x + kx))
            for z in gen_indices(iz, kz, sz) for y in gen_indices(iy, ky, sy) for x in gen_indices(ix, kx, sx)]


def predict_volume(model_fn, raw, patch_shape, stride_shape, halo, out_channels, save_segmentation=False):
    """model_fn: (1, 1, PD, PH, PW) float32 array -> (1, C, PD, PH, PW) float32 array.  raw: (D, H, W)."""
    padded = np.pad(raw, [(p, p) for p in halo], mode="reflect") if any(halo) else raw          # mirror_pad
    pmap = np.zeros((out_channels,) + raw.shape, dtype=np.float32)
    norm = np.zeros((out_channels,) + raw.shape, dtype=np.uint8)
    for idx in build_slices(raw.shape, patch_shape, stride_shape):
        pidx = tuple(slice(i.start, i.stop + 2 * h) for i, h in zip(idx, halo))                # hdf5.py:20-24
        pred = model_fn(padded[pidx][None, None])
        pred = pred[(..., *(slice(p, -p or None) for p in halo))]                               # remove_padding
        for p in pred:                                                                          # predictor.py:141-156
            index = (slice(0, out_channels),) + tuple(idx)
            pmap[index] += p
            norm[index] += 1
    result = pmap / norm                                                                        # predictor.py:164-168
    if save_segmentation:
        result = np.argmax(result, axis=0).astype("uint16")
    return result
