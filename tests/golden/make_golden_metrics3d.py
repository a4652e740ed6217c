"""G11: the 3-D validation metrics of the REAL reference (model/unet3d/metrics.py: MeanIoU, DiceCoefficient), build container only.

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_metrics3d.py

metrics.py imports skimage.metrics (absent third-party: empty stand-in, none of its names is called here) and `pytorch3dunet.unet3d.{losses,utils,
seg_metrics}` - the upstream package the reference's own model/unet3d/*.py files are copies of: they are aliased to those local files, as SURVEY.md §8c
does for `.se`."""
import importlib.util
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from _ref_import import _stub, import_reference  # noqa: E402


def load_ref_metrics():
    ns = import_reference()
    import model.unet3d.utils as ref_utils
    sk = _stub("skimage.metrics", adapted_rand_error=None, peak_signal_noise_ratio=None, mean_squared_error=None, contingency_table=None)
    sys.modules["skimage"].metrics = sk
    sys.modules["skimage.measure"].label = None
    sys.modules["pytorch3dunet.unet3d.losses"] = ns.losses3d
    sys.modules["pytorch3dunet.unet3d.utils"] = ref_utils
    for name in ("seg_metrics", "metrics"):
        spec = importlib.util.spec_from_file_location(f"pytorch3dunet.unet3d.{name}", f"/root/reference/model/unet3d/{name}.py")
        m = importlib.util.module_from_spec(spec)
        sys.modules[f"pytorch3dunet.unet3d.{name}"] = m
        spec.loader.exec_module(m)
    return sys.modules["pytorch3dunet.unet3d.metrics"]


def main():
    M = load_ref_metrics()
    rng = np.random.RandomState(11)
    N, C, D, H, W = 2, 3, 5, 6, 7
    logits = rng.randn(N, C, D, H, W).astype(np.float32)
    probs = np.exp(logits) / np.exp(logits).sum(1, keepdims=True)
    probs[0, :, 0, 0, :3] = 0.25                                  # exact ties: the first maximum wins
    probs[1, 1:, 1, 2, :] = probs[1, 1:, 1, 2, :].max()
    labels = rng.randint(0, C, size=(N, D, H, W)).astype(np.int64)
    onehot = np.eye(C, dtype=np.float32)[labels].transpose(0, 4, 1, 2, 3).copy()
    lab_ign = labels.copy()
    lab_ign[rng.rand(N, D, H, W) < 0.2] = -1
    oh_ign = np.eye(C, dtype=np.float32)[np.where(lab_ign < 0, 0, lab_ign)].transpose(0, 4, 1, 2, 3).copy()
    oh_ign[np.broadcast_to((lab_ign < 0)[:, None], oh_ign.shape)] = -1.0
    p1 = rng.rand(N, 1, D, H, W).astype(np.float32)
    p1[0, 0, 0, 0, :2] = 0.5                                      # not > 0.5
    t1 = (rng.rand(N, 1, D, H, W) > 0.5).astype(np.float32)
    out = {"probs": probs.astype(np.float32), "labels": labels, "onehot": onehot, "lab_ign": lab_ign, "oh_ign": oh_ign, "p1": p1, "t1": t1}
    T = torch.from_numpy
    out["miou_onehot"] = M.MeanIoU()(T(probs), T(onehot.copy())).numpy()
    out["miou_labels"] = M.MeanIoU()(T(probs), T(labels)).numpy()
    out["miou_skip0"] = M.MeanIoU(skip_channels=(0,))(T(probs), T(labels)).numpy()
    out["miou_lab_ign"] = M.MeanIoU(ignore_index=-1)(T(probs), T(lab_ign)).numpy()
    tmut = T(oh_ign.copy())
    out["miou_oh_ign"] = M.MeanIoU(ignore_index=-1)(T(probs), tmut).numpy()
    out["oh_ign_after"] = tmut.numpy()                             # the reference zeroes the ignored voxels of the caller's target
    out["miou_c1"] = M.MeanIoU()(T(p1), T(t1.copy())).numpy()
    out["dice"] = M.DiceCoefficient()(T(probs), T(onehot)).numpy()
    out["dice_c1"] = M.DiceCoefficient()(T(p1), T(t1)).numpy()
    np.savez_compressed(os.path.join(HERE, "g11_metrics3d.npz"), **out)
    print({k: float(v) for k, v in out.items() if v.ndim == 0})


if __name__ == "__main__":
    main()
