"""CPU restatement (numpy) of the reference's evaluation metrics, trainer/metrcis.py:61-109,153-168.  TEST INFRASTRUCTURE ONLY:
imported by tests/ (and nothing else).  Pinned by tests/golden/g6_metrics.npz, produced by the real `compute_metrics`."""
import numpy as np


def compute_iou(preds, labels, threshold=0.5):
    """metrcis.py:61-81"""
    preds = (preds > threshold).astype(np.float32)
    labels = (labels > threshold).astype(np.float32)
    inter = np.sum(preds * labels, axis=(1, 2))
    union = np.maximum(np.sum(preds, axis=(1, 2)) + np.sum(labels, axis=(1, 2)) - inter, 1e-6)
    return np.mean(inter / union)


def compute_dice(preds, labels, threshold=0.5):
    """metrcis.py:84-109 (epsilon enters the numerator once and each of the two sums once)"""
    preds = (preds > threshold).astype(np.float32)
    labels = (labels > threshold).astype(np.float32)
    inter = np.sum(preds * labels, axis=(1, 2))
    sp = np.sum(preds, axis=(1, 2)) + 1e-6
    sl = np.sum(labels, axis=(1, 2)) + 1e-6
    return np.mean((2.0 * inter + 1e-6) / (sp + sl))


def compute_metrics(logits, labels):
    """metrcis.py:153-168 without the matplotlib side effect"""
    preds = np.squeeze(logits, axis=1).astype(np.float32)
    labels = np.squeeze(labels, axis=1).astype(np.float32)
    preds = 1 / (1 + np.exp(-preds) + 1e-6)
    thr = np.mean(preds)
    return {"iou": compute_iou(preds, labels, thr), "dice": compute_dice(preds, labels, thr), "threshold": thr}


def mean_iou3d(probs, target, skip_channels=(), ignore_index=None):
    """model/unet3d/metrics.py:33-103 (+ expand_as_one_hot, model/unet3d/utils.py:222-254 for an integer label volume); float32 ratios and means
    like the torch scalars of the reference.  Pinned by tests/golden/g11_metrics3d.npz (real reference classes)."""
    N, C = probs.shape[:2]
    if target.ndim == 4:
        lab = target
        oh = np.stack([(lab == c) for c in range(C)], axis=1).astype(np.float32)
        if ignore_index is not None:
            oh[np.broadcast_to((lab == ignore_index)[:, None], oh.shape)] = ignore_index
        target = oh
    per_batch = []
    for n in range(N):
        p, t = probs[n], target[n].copy()
        if C == 1:
            pred = (p > 0.5).astype(np.uint8)
        else:
            pred = np.zeros_like(p, dtype=np.uint8)
            np.put_along_axis(pred, np.argmax(p, axis=0)[None], 1, axis=0)        # argmax = first maximum
        if ignore_index is not None:
            m = t == ignore_index
            pred[m] = 0
            t[m] = 0
        tb = t.astype(np.int64).astype(np.uint8)
        ious = [np.float32((pred[c] & tb[c]).sum()) / np.maximum(np.float32((pred[c] | tb[c]).sum()), np.float32(1e-8))
                for c in range(C) if c not in skip_channels]
        per_batch.append(np.mean(np.array(ious, dtype=np.float32), dtype=np.float32))
    return np.mean(np.array(per_batch, dtype=np.float32), dtype=np.float32)


def dice_coefficient3d(probs, target, eps=1e-6):
    """model/unet3d/metrics.py:15-30 over losses.py:7-33: mean_c 2*sum(p*t) / clamp(sum(p^2) + sum(t^2), eps), channels flattened over (N, D, H, W)"""
    C = probs.shape[1]
    p = np.moveaxis(probs, 1, 0).reshape(C, -1).astype(np.float64)
    t = np.moveaxis(target, 1, 0).reshape(C, -1).astype(np.float64)
    d = 2 * (p * t).sum(1) / np.maximum((p * p).sum(1) + (t * t).sum(1), eps)
    return d.mean()
