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
