"""G6: the reference's `compute_metrics` (trainer/metrcis.py:153-168) on fixed logits / labels.

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_metrics.py

Runs the real function (its matplotlib side effect writes output.png into a temporary working directory)."""
import os
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from _ref_import import import_reference  # noqa: E402


class EP:
    def __init__(self, predictions, label_ids):
        self.predictions, self.label_ids = predictions, label_ids


def main():
    import matplotlib
    matplotlib.use("Agg")
    import_reference()
    import trainer.metrcis as M          # the reference's module (sys.path set by import_reference)
    rng = np.random.RandomState(6)
    out = {}
    cases = {"a": (5, 48, 40), "b": (3, 64, 64), "c": (2, 16, 16)}
    cwd = os.getcwd()
    with tempfile.TemporaryDirectory() as tmp:
        os.chdir(tmp)
        try:
            for tag, (n, h, w) in cases.items():
                labels = (rng.rand(n, 1, h, w) > 0.6).astype(np.float32)
                logits = (rng.randn(n, 1, h, w) * 1.5 + (labels * 2 - 1) * 1.2).astype(np.float32)
                if tag == "c":
                    labels[1] = 0          # an empty label map: exercises the epsilons
                    logits[1] = -30.0
                r = M.compute_metrics(EP(logits, labels))
                out[f"{tag}_logits"], out[f"{tag}_labels"] = logits, labels
                out[f"{tag}_iou"], out[f"{tag}_dice"] = np.float64(r["iou"]), np.float64(r["dice"])
                out[f"{tag}_iou05"] = np.float64(M.compute_iou(logits[:, 0], labels[:, 0], 0.5))
                out[f"{tag}_dice05"] = np.float64(M.compute_dice(logits[:, 0], labels[:, 0], 0.5))
        finally:
            os.chdir(cwd)
    np.savez_compressed(os.path.join(HERE, "g6_metrics.npz"), **out)
    print("wrote g6_metrics.npz", {k: float(v) for k, v in out.items() if v.ndim == 0})


if __name__ == "__main__":
    main()
