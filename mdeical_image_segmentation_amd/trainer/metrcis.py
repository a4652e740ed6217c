"""Mirror of the reference's trainer/metrcis.py (sic): `compute_metrics` (:153-168) and its helpers `compute_iou` (:61-81) /
`compute_dice` (:84-109), evaluated on the MI355X by `mis_seg_metrics` (csrc/metrics.hip): the reference's sigmoid with +1e-6 in the
denominator, threshold = global mean probability, per-sample IoU / Dice, mean over samples.

Inputs may be numpy arrays (what HF Trainer's EvalPrediction carries; uploaded once) or CUDA tensors (an evaluation loop that
never leaves the device).  Returns numpy float32 scalars like the reference.  The reference's side effect of writing
`output.png` (matplotlib) is not reproduced.  No host fallback: without the HIP library the call raises."""
import numpy as np
import torch

from .. import ops
from .._lib import MisError, check, load, stream_ptr


def _dev32(a):
    if isinstance(a, np.ndarray):
        a = torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32))
    if not isinstance(a, torch.Tensor):
        raise MisError(f"expected a numpy array or a torch tensor, got {type(a)}")
    if a.device.type != "cuda":
        a = a.cuda()
    return a.to(torch.float32).contiguous()


def _seg_metrics(values, labels, is_logits, auto_thr, thr):
    v, l = _dev32(values), _dev32(labels)
    if v.shape != l.shape or v.dim() < 2:
        raise MisError(f"predictions {tuple(v.shape)} and labels {tuple(l.shape)} must have the same (N, ...) shape")
    N = v.shape[0]
    npix = v[0].numel()
    lib = load()
    ws = ops.workspace(lib.mis_seg_metrics_workspace_bytes(N, npix), v.device, "metrics")
    out = torch.empty(3, dtype=torch.float32, device=v.device)
    check(lib.mis_seg_metrics(v.data_ptr(), l.data_ptr(), N, npix, 1 if is_logits else 0, 1 if auto_thr else 0, float(thr), ws.data_ptr(),
                              out.data_ptr(), stream_ptr()), "mis_seg_metrics")
    return out.cpu().numpy()


def compute_iou(preds, labels, threshold=0.5):
    """(N, H, W) predictions / labels, both binarised with `threshold`; mean over samples of |A & B| / max(|A u B|, 1e-6)."""
    return _seg_metrics(preds, labels, False, False, threshold)[0]


def compute_dice(preds, labels, threshold=0.5):
    return _seg_metrics(preds, labels, False, False, threshold)[1]


def compute_metrics(p):
    """p: EvalPrediction-like with .predictions (N, 1, H, W) logits and .label_ids (N, 1, H, W)."""
    logits, labels = p.predictions, p.label_ids
    if logits.shape[1] != 1 or labels.shape[1] != 1:
        raise MisError("compute_metrics squeezes a singleton channel axis (reference metrcis.py:156-157): expected (N, 1, H, W)")
    r = _seg_metrics(logits, labels, True, True, 0.0)
    return {"iou": r[0], "dice": r[1]}
