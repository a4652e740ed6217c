"""Mirror of the reference's patch-tiled predictor (model/unet3d/predictor.py:64-168 `StandardPredictor`) with the tiling helpers it
relies on (dataset/unet3d_dataset/utils.py: `SliceBuilder._gen_indices` :110-116, `mirror_pad` :314-342, `remove_padding` :345-361),
run entirely on the MI355X: the raw volume stays in HBM, patches (with their mirrored halo) are gathered from it by index arithmetic,
the network is the fused engine, and the halo removal / accumulation / visit counting / normalisation / arg-max are three small
kernels (csrc/predictor.hip).  Patches are accumulated in the reference's order (z, y, x raster), one launch each, so overlapping
regions are summed in the same order as the reference's numpy `+=`.

HDF5 input / output (`__call__(test_loader)`, `LazyPredictor`, `DSB2018Predictor`) is outside the hot path: `predict_volume` takes and
returns tensors."""
import numpy as np
import torch

from ... import ops
from ..._lib import MisError, check, load, stream_ptr


def gen_indices(i, k, s):
    """SliceBuilder._gen_indices (dataset/unet3d_dataset/utils.py:110-116): starts 0, s, 2s, ... plus a last patch flush with the end."""
    if i < k:
        raise MisError("Sample size has to be bigger than the patch size")
    j = 0
    for j in range(0, i - k + 1, s):
        yield j
    if j + k < i:
        yield i - k


def build_origins(volume_shape, patch_shape, stride_shape):
    """SliceBuilder._build_slices (:85-108) reduced to the patch start positions, z-major raster order."""
    (iz, iy, ix), (kz, ky, kx), (sz, sy, sx) = volume_shape, patch_shape, stride_shape
    return [(z, y, x) for z in gen_indices(iz, kz, sz) for y in gen_indices(iy, ky, sy) for x in gen_indices(ix, kx, sx)]


class StandardPredictor:
    def __init__(self, model, output_dir=None, out_channels=None, output_dataset="predictions", save_segmentation=False,
                 prediction_channel=None, **kwargs):
        if out_channels is None:
            raise MisError("StandardPredictor needs out_channels")
        self.model = model
        self.output_dir = output_dir
        self.out_channels = out_channels
        self.output_dataset = output_dataset
        self.save_segmentation = save_segmentation
        self.prediction_channel = prediction_channel

    def __call__(self, test_loader):
        raise NotImplementedError("HDF5 loaders / writers are outside the accelerated hot path: use predict_volume(raw, patch_shape, "
                                  "stride_shape, halo_shape) with the volume as a tensor")

    @torch.no_grad()
    def predict_volume(self, raw, patch_shape, stride_shape, halo_shape=(0, 0, 0), batch_size=1, activation="none"):
        """raw: (D, H, W) or (C, D, H, W) fp32 volume (numpy or tensor).  Returns a CUDA tensor: the averaged prediction map
        (out_channels or 1, D, H, W) fp32, or, with save_segmentation, the channel arg-max (D, H, W) as uint16 (torch.uint16)."""
        if isinstance(raw, np.ndarray):
            raw = torch.from_numpy(np.ascontiguousarray(raw, dtype=np.float32))
        vol = raw.to(device="cuda", dtype=torch.float32).contiguous()
        if vol.dim() == 3:
            vol = vol.unsqueeze(0)
        if vol.dim() != 4:
            raise MisError("raw must be (D, H, W) or (C, D, H, W)")
        C, D, H, W = vol.shape
        kd, kh, kw = (int(v) for v in patch_shape)
        hd, hh, hw = (int(v) for v in halo_shape)
        act = {"none": 0, "sigmoid": 1, "softmax": 2}[activation]
        origins = build_origins((D, H, W), (kd, kh, kw), tuple(int(v) for v in stride_shape))
        PD, PH, PW = kd + 2 * hd, kh + 2 * hh, kw + 2 * hw
        lib = load()
        dev = vol.device
        out_c = 1 if self.prediction_channel is not None else self.out_channels
        pmap = torch.zeros(out_c, D, H, W, dtype=torch.float32, device=dev)
        norm = torch.zeros(D, H, W, dtype=torch.uint8, device=dev)
        org_dev = torch.tensor(origins, dtype=torch.int32, device=dev)
        self.model.eval()
        for b0 in range(0, len(origins), batch_size):
            nb = min(batch_size, len(origins) - b0)
            patches = torch.empty(nb, C, PD, PH, PW, dtype=torch.float32, device=dev)
            check(lib.mis_patch_gather_reflect(vol.data_ptr(), C, D, H, W, org_dev[b0:b0 + nb].data_ptr(), nb, PD, PH, PW, hd, hh, hw,
                                               patches.data_ptr(), stream_ptr()), "mis_patch_gather_reflect")
            pred = self.model(patches)
            if pred.dtype != torch.float32 or not pred.is_contiguous() or tuple(pred.shape) != (nb, self.out_channels, PD, PH, PW):
                raise MisError(f"the model returned {tuple(pred.shape)} {pred.dtype}; expected fp32 {(nb, self.out_channels, PD, PH, PW)}")
            for i in range(nb):
                oz, oy, ox = origins[b0 + i]
                check(lib.mis_patch_accumulate(pred[i].data_ptr(), self.out_channels, PD, PH, PW, hd, hh, hw, act,
                                               -1 if self.prediction_channel is None else int(self.prediction_channel), oz, oy, ox,
                                               pmap.data_ptr(), norm.data_ptr(), D, H, W, stream_ptr()), "mis_patch_accumulate")
        nvox = D * H * W
        if self.save_segmentation:
            seg = torch.empty(D, H, W, dtype=torch.uint16, device=dev)
            check(lib.mis_pred_finalize(pmap.data_ptr(), norm.data_ptr(), out_c, nvox, None, seg.data_ptr(), stream_ptr()), "mis_pred_finalize")
            return seg
        prob = torch.empty_like(pmap)
        check(lib.mis_pred_finalize(pmap.data_ptr(), norm.data_ptr(), out_c, nvox, prob.data_ptr(), None, stream_ptr()), "mis_pred_finalize")
        return prob
