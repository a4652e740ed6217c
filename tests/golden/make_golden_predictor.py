"""G8: patch-tiled prediction with the reference's own building blocks.

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_predictor.py

`model/unet3d/predictor.py` itself needs h5py and the pytorch3dunet package (absent), so its loop (:111-168) is replayed here on the
REAL `SliceBuilder._build_slices`, `mirror_pad`, `remove_padding` (dataset/unet3d_dataset/utils.py, loaded from the file) and the REAL
`UNet3D` (eval mode).  Stores the volume, the averaged logits map (statistics + strided sample), and the arg-max segmentation."""
import importlib.util
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from _ref_import import import_reference  # noqa: E402

torch.set_num_threads(8)


def main():
    ns = import_reference()
    spec = importlib.util.spec_from_file_location("ref_ds_utils", "/root/reference/dataset/unet3d_dataset/utils.py")
    U = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(U)
    torch.manual_seed(0)
    net = ns.model3d.UNet3D(1, 3, f_maps=[64, 128], num_levels=2).eval()
    rng = np.random.RandomState(8)
    raw = rng.randn(24, 80, 72).astype(np.float32)
    patch, stride, halo = (8, 64, 64), (8, 48, 40), (2, 4, 4)
    slices = U.SliceBuilder._build_slices(raw, patch, stride)
    padded = U.mirror_pad(raw, halo)
    pmap = np.zeros((3,) + raw.shape, dtype="float32")
    norm = np.zeros((3,) + raw.shape, dtype="uint8")
    with torch.no_grad():
        for idx in slices:
            pidx = tuple(slice(i.start, i.stop + 2 * h) for i, h in zip(idx, halo))       # hdf5.py:_create_padded_indexes
            pred = net(torch.from_numpy(padded[pidx][None, None]))
            pred = U.remove_padding(pred, halo).cpu().numpy()
            for p in pred:
                index = (slice(0, 3),) + tuple(idx)
                pmap[index] += p
                norm[index] += 1
    result = pmap / norm
    seg = np.argmax(result, axis=0).astype("uint16")
    flat = result.reshape(-1)
    sample_idx = np.linspace(0, flat.size - 1, 4096).astype(np.int64)
    out = {"raw": raw, "patch": np.array(patch), "stride": np.array(stride), "halo": np.array(halo),
           "origins": np.array([[s.start for s in idx] for idx in slices]), "norm_max": np.array(norm.max()),
           "seg": seg, "sample_idx": sample_idx, "sample": flat[sample_idx],
           "stats": np.array([flat.astype(np.float64).sum(), np.abs(flat).astype(np.float64).sum()])}
    np.savez_compressed(os.path.join(HERE, "g8_predictor.npz"), **out)
    print("wrote g8_predictor.npz", sum(a.nbytes for a in out.values()) // 1024, "KiB;", len(slices), "patches; max visits", norm.max())


if __name__ == "__main__":
    main()
