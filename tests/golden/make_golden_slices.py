"""G14: patch positions of the REAL reference slice builders (dataset/unet3d_dataset/utils.py SliceBuilder / FilterSliceBuilder), build container only.

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_slices.py

utils.py is loaded from its file (the `dataset` package itself needs albumentations); h5py and friends are the usual stand-ins."""
import importlib.util
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from _ref_import import import_reference  # noqa: E402


def starts(slices):
    return np.array([[s.start for s in idx] + [s.stop for s in idx] for idx in slices], dtype=np.int64)


def main():
    import_reference()
    spec = importlib.util.spec_from_file_location("ref_ds_utils", "/root/reference/dataset/unet3d_dataset/utils.py")
    U = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(U)
    rng = np.random.RandomState(14)
    out = {}
    raw3 = rng.rand(30, 100, 90).astype(np.float32)
    raw4 = rng.rand(2, 17, 70, 131).astype(np.float32)
    lab3 = (rng.rand(30, 100, 90) > 0.45).astype(np.int64)
    lab3[:, :40] = 0
    lab3[rng.rand(30, 100, 90) < 0.1] = -1
    out["lab3"] = lab3.astype(np.int8)
    sb = U.SliceBuilder(raw3, lab3, None, (8, 64, 64), (4, 32, 40))
    out["sb3_raw"], out["sb3_label"] = starts(sb.raw_slices), starts(sb.label_slices)
    sb = U.SliceBuilder(raw4, None, None, (17, 64, 64), (17, 64, 64))
    out["sb4_raw"] = starts(sb.raw_slices)
    sb = U.SliceBuilder(raw3, None, None, (5, 7, 9), (5, 6, 4), skip_shape_check=True)
    out["sb_small"] = starts(sb.raw_slices)
    for tag, kw in (("a", dict(threshold=0.3, slack_acceptance=0.2)), ("b", dict(ignore_index=-1, threshold=0.25, slack_acceptance=0.05))):
        fb = U.FilterSliceBuilder(raw3, lab3, None, (8, 64, 64), (4, 32, 40), **kw)
        out[f"fb_{tag}_raw"], out[f"fb_{tag}_label"] = starts(fb.raw_slices), starts(fb.label_slices)
    out["stats_in"] = rng.randn(3, 9, 11, 13).astype(np.float32)
    st = U.calculate_stats(out["stats_in"])
    out["stats"] = np.array([st["pmin"], st["pmax"], st["mean"], st["std"]], dtype=np.float64)
    np.savez_compressed(os.path.join(HERE, "g14_slices.npz"), **out)
    print({k: v.shape for k, v in out.items()})


if __name__ == "__main__":
    main()
