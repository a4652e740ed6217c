"""Mirror of the hot-path part of model/unet3d/losses.py (:7-33, :83-129, :167-178, :258-306) as stand-alone torch
modules (compatibility surface for an EXTERNAL loss on the logits of the fused UNet3D; the fused train step computes
BCE+Dice inside the head kernel instead - engine3d.UNet3DEngine)."""
import torch
from torch import nn


def flatten(tensor):
    C = tensor.size(1)
    axis_order = (1, 0) + tuple(range(2, tensor.dim()))
    return tensor.permute(axis_order).contiguous().view(C, -1)


def compute_per_channel_dice(input, target, epsilon=1e-6, weight=None):
    assert input.size() == target.size(), "'input' and 'target' must have the same shape"
    input = flatten(input)
    target = flatten(target).float()
    intersect = (input * target).sum(-1)
    if weight is not None:
        intersect = weight * intersect
    denominator = (input * input).sum(-1) + (target * target).sum(-1)
    return 2 * (intersect / denominator.clamp(min=epsilon))


class _AbstractDiceLoss(nn.Module):
    def __init__(self, weight=None, normalization="sigmoid"):
        super().__init__()
        self.register_buffer("weight", weight)
        assert normalization in ["sigmoid", "softmax", "none"]
        if normalization == "sigmoid":
            self.normalization = nn.Sigmoid()
        elif normalization == "softmax":
            self.normalization = nn.Softmax(dim=1)
        else:
            self.normalization = lambda x: x

    def dice(self, input, target, weight):
        raise NotImplementedError

    def forward(self, input, target):
        input = self.normalization(input)
        return 1. - torch.mean(self.dice(input, target, weight=self.weight))


class DiceLoss(_AbstractDiceLoss):
    def dice(self, input, target, weight):
        return compute_per_channel_dice(input, target, weight=self.weight)


class BCEDiceLoss(nn.Module):
    def __init__(self, alpha, beta):
        super().__init__()
        self.alpha = alpha
        self.bce = nn.BCEWithLogitsLoss()
        self.beta = beta
        self.dice = DiceLoss()

    def forward(self, input, target):
        return self.alpha * self.bce(input, target) + self.beta * self.dice(input, target)


def get_loss_criterion(config):
    """losses.py:273-306 (mutates config['loss'] via pop, like the reference); BCEDiceLoss / DiceLoss / BCEWithLogitsLoss /
    CrossEntropyLoss are built, the rest of the factory is out of scope."""
    assert "loss" in config, "Could not find loss function configuration"
    loss_config = config["loss"]
    name = loss_config.pop("name")
    loss_config.pop("ignore_index", None)
    loss_config.pop("skip_last_target", False)
    loss_config.pop("weight", None)
    loss_config.pop("pos_weight", None)
    if name == "BCEDiceLoss":
        loss = BCEDiceLoss(loss_config.get("alpha", 1.), loss_config.get("beta", 1.))
    elif name == "DiceLoss":
        loss = DiceLoss(normalization=loss_config.get("normalization", "sigmoid"))
    elif name == "BCEWithLogitsLoss":
        loss = nn.BCEWithLogitsLoss()
    elif name == "CrossEntropyLoss":
        loss = nn.CrossEntropyLoss()
    else:
        raise NotImplementedError(f"Unsupported loss function on the accelerated path: '{name}'")
    if torch.cuda.is_available():
        loss = loss.cuda()
    return loss
