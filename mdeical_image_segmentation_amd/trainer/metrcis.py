"""Mirror of trainer/metrcis.py (sic) :61-109,153-168: eval-time IoU / Dice on host numpy.
Same formulas (sigmoid with +1e-6 in the denominator, threshold = global mean of the probabilities);
the reference's side effect of writing output.png (matplotlib) is not reproduced."""
import numpy as np


def compute_iou(preds, labels, threshold=0.5):
    preds = (preds > threshold).astype(np.float32)
    labels = (labels > threshold).astype(np.float32)
    intersection = np.sum(preds * labels, axis=(1, 2))
    union = np.sum(preds, axis=(1, 2)) + np.sum(labels, axis=(1, 2)) - intersection
    union = np.maximum(union, 1e-6)
    return np.mean(intersection / union)


def compute_dice(preds, labels, threshold=0.5):
    preds = (preds > threshold).astype(np.float32)
    labels = (labels > threshold).astype(np.float32)
    intersection = np.sum(preds * labels, axis=(1, 2))
    sum_pred = np.sum(preds, axis=(1, 2)) + 1e-6
    sum_lab = np.sum(labels, axis=(1, 2)) + 1e-6
    return np.mean(((2.0 * intersection) + 1e-6) / (sum_pred + sum_lab))


def compute_metrics(p):
    logits, labels = p.predictions, p.label_ids
    preds = np.squeeze(logits, axis=1).astype(np.float32)
    labels = np.squeeze(labels, axis=1).astype(np.float32)
    preds = 1 / (1 + np.exp(-preds) + 1e-6)
    threshold = np.mean(preds)
    return {"iou": compute_iou(preds, labels, threshold), "dice": compute_dice(preds, labels, threshold)}
