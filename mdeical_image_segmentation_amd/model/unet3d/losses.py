"""Mirror of the hot-path part of the reference's model/unet3d/losses.py (`flatten` :258-270, `compute_per_channel_dice` :7-33,
`_AbstractDiceLoss` :83-116, `DiceLoss` :119-129, `BCEDiceLoss` :167-178, `get_loss_criterion` :273-306) as nn.Modules whose forward and
backward are the stand-alone HIP loss kernels (csrc/losses.hip, `mis_bcedice_fwd/_bwd`): two reduction passes + one elementwise pass.

This is the surface for an EXTERNAL criterion applied to the logits of the fused UNet3D (and for the HF wrapper's double-sigmoid quirk);
the fused train step computes the same BCE+Dice inside the 1x1-head kernel instead (engine3d.UNet3DEngine).  CUDA tensors only."""
import torch
from torch import nn

from ... import ops
from ..._lib import MisError, check, load, stream_ptr


def flatten(tensor):
    """(N, C, D, H, W) -> (C, N * D * H * W)   [tensor plumbing, losses.py:258-270]"""
    C = tensor.size(1)
    axis_order = (1, 0) + tuple(range(2, tensor.dim()))
    return tensor.permute(axis_order).contiguous().view(C, -1)


def _prep(input, target):
    if input.size() != target.size():
        raise MisError("'input' and 'target' must have the same shape")
    if input.device.type != "cuda":
        raise MisError(f"the loss kernels run on MI355X only: got a tensor on {input.device} (no CPU fallback)")
    x = input.contiguous().float()
    t = target.to(device=x.device).contiguous().float()
    N, C = x.shape[0], x.shape[1]
    S = x[0, 0].numel()
    return x, t, N, C, S


class _BCEDice(torch.autograd.Function):
    @staticmethod
    def forward(ctx, input, target, alpha, beta, normalize):
        x, t, N, C, S = _prep(input, target)
        lib = load()
        ws = ops.workspace(lib.mis_bcedice_workspace_bytes(C), x.device, "bcedice")
        out = torch.empty(2 + 4 * C, dtype=torch.float32, device=x.device)
        check(lib.mis_bcedice_fwd(x.data_ptr(), t.data_ptr(), N, C, S, float(alpha), float(beta), 1 if normalize else 0, ws.data_ptr(),
                                  out.data_ptr(), stream_ptr()), "mis_bcedice_fwd")
        ctx.save_for_backward(x, t, out)
        ctx.cfg = (N, C, S, float(alpha), float(beta), 1 if normalize else 0, input.shape, input.dtype)
        return out[0].clone(), out[2:].view(C, 4)[:, 3].clone()

    @staticmethod
    def backward(ctx, g, g_dice):
        x, t, out = ctx.saved_tensors
        N, C, S, alpha, beta, normalize, shape, dtype = ctx.cfg
        dx = torch.empty_like(x)
        gg = g.contiguous().float().reshape(1)
        check(load().mis_bcedice_bwd(x.data_ptr(), t.data_ptr(), N, C, S, alpha, beta, normalize, out.data_ptr(), gg.data_ptr(), dx.data_ptr(),
                                     stream_ptr()), "mis_bcedice_bwd")
        return dx.view(shape).to(dtype), None, None, None, None


def compute_per_channel_dice(input, target, epsilon=1e-6, weight=None):
    """per-channel Dice of already normalised probabilities: 2 * sum(p*t) / clamp(sum(p^2) + sum(t^2), eps)   [losses.py:7-33]"""
    if weight is not None or epsilon != 1e-6:
        raise NotImplementedError("compute_per_channel_dice on MI355X: weight=None, epsilon=1e-6 (the reference's defaults) are built")
    return _BCEDice.apply(input, target, 0.0, 1.0, False)[1]


class _AbstractDiceLoss(nn.Module):
    def __init__(self, weight=None, normalization="sigmoid"):
        super().__init__()
        self.register_buffer("weight", weight)
        assert normalization in ["sigmoid", "softmax", "none"]
        if normalization == "softmax" or weight is not None:
            raise NotImplementedError("Dice loss on MI355X: normalization 'sigmoid' (default) or 'none', weight=None are built")
        self.normalization_name = normalization
        self.normalization = nn.Sigmoid() if normalization == "sigmoid" else (lambda x: x)     # attribute kept for interface parity

    def dice(self, input, target, weight):
        raise NotImplementedError

    def forward(self, input, target):
        # 1 - mean_c dice_c with the normalisation fused into the kernel
        return _BCEDice.apply(input, target, 0.0, 1.0, self.normalization_name == "sigmoid")[0]


class DiceLoss(_AbstractDiceLoss):
    def dice(self, input, target, weight):
        return compute_per_channel_dice(input, target, weight=self.weight)


class BCEDiceLoss(nn.Module):
    """alpha * BCEWithLogits + beta * Dice (sigmoid normalisation), one fused forward / backward"""

    def __init__(self, alpha, beta):
        super().__init__()
        self.alpha = alpha
        self.bce = nn.BCEWithLogitsLoss()      # attributes kept for interface parity (reference :172-175)
        self.beta = beta
        self.dice = DiceLoss()

    def forward(self, input, target):
        return _BCEDice.apply(input, target, self.alpha, self.beta, True)[0]


class _BCEOnly(nn.Module):
    def forward(self, input, target):
        return _BCEDice.apply(input, target, 1.0, 0.0, True)[0]


class _CE3d(torch.autograd.Function):
    @staticmethod
    def forward(ctx, input, target, ignore_index):
        if input.device.type != "cuda":
            raise MisError(f"the loss kernels run on MI355X only: got a tensor on {input.device} (no CPU fallback)")
        x = input.contiguous().float()
        lab = target.to(device=x.device, dtype=torch.int64).contiguous()
        N, C = x.shape[0], x.shape[1]
        S = x[0, 0].numel()
        if tuple(lab.shape) != (N,) + tuple(x.shape[2:]):
            raise MisError(f"CrossEntropyLoss: labels {tuple(lab.shape)} do not match logits {tuple(x.shape)}")
        lib = load()
        ws = ops.workspace(lib.mis_loss_workspace_bytes(), x.device, "loss")
        out = torch.empty(2, dtype=torch.float32, device=x.device)
        check(lib.mis_ce3d_fwd(x.data_ptr(), lab.data_ptr(), N, C, S, int(ignore_index), ws.data_ptr(), out.data_ptr(), stream_ptr()), "mis_ce3d_fwd")
        ctx.save_for_backward(x, lab, out)
        ctx.cfg = (N, C, S, int(ignore_index), input.shape, input.dtype)
        return out[0].clone()

    @staticmethod
    def backward(ctx, g):
        x, lab, out = ctx.saved_tensors
        N, C, S, ign, shape, dtype = ctx.cfg
        dx = torch.empty_like(x)
        gg = g.contiguous().float().reshape(1)
        check(load().mis_ce3d_bwd(x.data_ptr(), lab.data_ptr(), N, C, S, ign, out.data_ptr(), gg.data_ptr(), dx.data_ptr(), stream_ptr()), "mis_ce3d_bwd")
        return dx.view(shape).to(dtype), None, None


class CrossEntropyLoss(nn.Module):
    """nn.CrossEntropyLoss(weight=None, ignore_index) on (N, C, D, H, W) logits and (N, D, H, W) integer labels (losses.py:354-356)"""

    def __init__(self, weight=None, ignore_index=-100):
        super().__init__()
        if weight is not None:
            raise NotImplementedError("CrossEntropyLoss on MI355X: class weights are not built")
        self.ignore_index = ignore_index

    def forward(self, input, target):
        return _CE3d.apply(input, target, self.ignore_index)


class _PointLoss(torch.autograd.Function):
    @staticmethod
    def forward(ctx, input, target, kind):
        x, t, _, _, _ = _prep(input, target)
        lib = load()
        ws = ops.workspace(lib.mis_loss_workspace_bytes(), x.device, "loss")
        out = torch.empty(2, dtype=torch.float32, device=x.device)
        check(lib.mis_pointloss_fwd(kind, x.data_ptr(), t.data_ptr(), x.numel(), ws.data_ptr(), out.data_ptr(), stream_ptr()), "mis_pointloss_fwd")
        ctx.save_for_backward(x, t)
        ctx.cfg = (kind, input.shape, input.dtype)
        return out[0].clone()

    @staticmethod
    def backward(ctx, g):
        x, t = ctx.saved_tensors
        kind, shape, dtype = ctx.cfg
        dx = torch.empty_like(x)
        gg = g.contiguous().float().reshape(1)
        check(load().mis_pointloss_bwd(kind, x.data_ptr(), t.data_ptr(), x.numel(), gg.data_ptr(), dx.data_ptr(), stream_ptr()), "mis_pointloss_bwd")
        return dx.view(shape).to(dtype), None, None


class MSELoss(nn.Module):
    def forward(self, input, target):
        return _PointLoss.apply(input, target, 0)


class L1Loss(nn.Module):
    def forward(self, input, target):
        return _PointLoss.apply(input, target, 1)


class SmoothL1Loss(nn.Module):
    def forward(self, input, target):
        return _PointLoss.apply(input, target, 2)


def get_loss_criterion(config):
    """losses.py:273-306 (mutates config['loss'] via pop, like the reference).  Built on the HIP loss kernels: BCEDiceLoss, DiceLoss, BCEWithLogitsLoss,
    CrossEntropyLoss (ignore_index), MSELoss, L1Loss, SmoothL1Loss; the weighted / pixel-wise / generalized variants and the masking wrappers raise."""
    assert "loss" in config, "Could not find loss function configuration"
    loss_config = config["loss"]
    name = loss_config.pop("name")
    ignore_index = loss_config.pop("ignore_index", None)
    skip_last_target = loss_config.pop("skip_last_target", False)
    weight = loss_config.pop("weight", None)
    pos_weight = loss_config.pop("pos_weight", None)
    if weight is not None or pos_weight is not None or skip_last_target or (ignore_index is not None and name != "CrossEntropyLoss"):
        raise NotImplementedError(f"loss '{name}': weight / pos_weight / skip_last_target / ignore_index masking are not built on the accelerated path")
    if name == "BCEDiceLoss":
        loss = BCEDiceLoss(loss_config.get("alpha", 1.), loss_config.get("beta", 1.))
    elif name == "DiceLoss":
        loss = DiceLoss(normalization=loss_config.get("normalization", "sigmoid"))
    elif name == "BCEWithLogitsLoss":
        loss = _BCEOnly()
    elif name == "CrossEntropyLoss":
        loss = CrossEntropyLoss(ignore_index=ignore_index if ignore_index is not None else -100)
    elif name in ("MSELoss", "L1Loss", "SmoothL1Loss"):
        loss = {"MSELoss": MSELoss, "L1Loss": L1Loss, "SmoothL1Loss": SmoothL1Loss}[name]()
    else:
        raise NotImplementedError(f"Unsupported loss function on the accelerated path: '{name}'")
    if torch.cuda.is_available():
        loss = loss.cuda()
    return loss
