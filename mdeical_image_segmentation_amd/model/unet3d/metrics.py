"""Mirror of the evaluation criteria of the reference's 3-D validation loop (model/unet3d/metrics.py): `MeanIoU` (:33-103, with
`_binarize_predictions` = first-maximum one-hot / > 0.5 for one channel, and `expand_as_one_hot` of model/unet3d/utils.py:222-254 for label
targets) and `DiceCoefficient` (:15-30), evaluated on the MI355X: the per-(sample, channel) intersection / union COUNTS come from one pass of
`mis_iou3d_counts` (csrc/metrics.hip, exact integers); the handful of ratios and means that follow are computed from them with the reference's own
expression, so the result is the same CPU float32 scalar tensor the reference returns.  The segmentation-instance metrics that need skimage
(AdaptedRandError, AveragePrecision, ...) are out of scope.  CUDA tensors only: no host fallback."""
import torch

from ... import ops
from ..._lib import MisError, check, load, stream_ptr
from .losses import compute_per_channel_dice


class DiceCoefficient:
    """metrics.py:15-30: mean over channels of the per-channel Dice of probability maps."""

    def __init__(self, epsilon=1e-6, **kwargs):
        self.epsilon = epsilon

    def __call__(self, input, target):
        return torch.mean(compute_per_channel_dice(input, target, epsilon=self.epsilon))


class MeanIoU:
    """metrics.py:33-103."""

    def __init__(self, skip_channels=(), ignore_index=None, **kwargs):
        self.ignore_index = ignore_index
        self.skip_channels = skip_channels

    def __call__(self, input, target):
        assert input.dim() == 5
        if input.device.type != "cuda" or target.device.type != "cuda":
            raise MisError("MeanIoU runs on MI355X only: got tensors on %s / %s (no CPU fallback)" % (input.device, target.device))
        N, C = input.shape[0], input.shape[1]
        S = input[0, 0].numel()
        labels = target.dim() == 4
        if labels:
            assert tuple(target.shape) == (N,) + tuple(input.shape[2:])
            t = target.to(torch.int64).contiguous()
        else:
            assert input.size() == target.size()
            t = target.to(torch.float32).contiguous()
        p = input.to(torch.float32).contiguous()
        counts = torch.empty(N, C, 2, dtype=torch.int64, device=p.device)
        check(load().mis_iou3d_counts(p.data_ptr(), t.data_ptr(), 1 if labels else 0, N, C, S, 0 if self.ignore_index is None else 1,
                                      0 if self.ignore_index is None else int(self.ignore_index), counts.data_ptr(), stream_ptr()), "mis_iou3d_counts")
        if not labels and self.ignore_index is not None and target.dtype == torch.float32 and target.is_contiguous():
            target[target == self.ignore_index] = 0          # the reference zeroes the ignored voxels of the caller's target in place (:66-68)
        counts = counts.cpu()
        per_batch_iou = []
        for n in range(N):
            per_channel_iou = []
            for c in range(C):
                if c in self.skip_channels:
                    continue
                per_channel_iou.append(counts[n, c, 0].float() / torch.clamp(counts[n, c, 1].float(), min=1e-8))
            assert per_channel_iou, "All channels were ignored from the computation"
            per_batch_iou.append(torch.mean(torch.tensor(per_channel_iou)))
        return torch.mean(torch.tensor(per_batch_iou))


def get_evaluation_metric(config):
    """metrics.py:430-445: config['eval_metric'] = {'name': 'MeanIoU' | 'DiceCoefficient', ...}"""
    assert 'eval_metric' in config, 'Could not find evaluation metric configuration'
    metric_config = config['eval_metric']
    classes = {"MeanIoU": MeanIoU, "DiceCoefficient": DiceCoefficient}
    if metric_config['name'] not in classes:
        raise NotImplementedError(f"eval metric {metric_config['name']}: only MeanIoU and DiceCoefficient are built on MI355X (SURVEY.md §8f4)")
    return classes[metric_config['name']](**metric_config)
