"""Mirror of the reference's model/unet2d/loss.py (:21-70): `MSSSIMLoss`, `IoULoss`, `F1Loss`, `SegmentationLoss` - the criterion of the
UNet 3+ models - with forward and backward on the HIP kernels of csrc/segloss.hip (`mis_segloss_fwd/_bwd`).

The MS-SSIM term restates pytorch_msssim 1.0.0 `MS_SSIM(data_range=1.0, size_average=True, channel=1)` (third-party, not shipped with the
reference): 11-tap Gaussian window (sigma 1.5), five scales, weights [0.0448, 0.2856, 0.3001, 0.2363, 0.1333].  Inputs: logits and
targets of shape (N, 1, H, W) with min(H, W) > 160 (the package's own requirement; F1Loss / IoULoss on their own take any size, as the reference's do).  CUDA tensors only."""
import torch
from torch import nn

from ... import ops
from ..._lib import MisError, check, load, stream_ptr


class _SegLoss(torch.autograd.Function):
    @staticmethod
    def forward(ctx, inputs, targets, w_f1, w_ms, w_iou):
        if inputs.device.type != "cuda":
            raise MisError(f"the loss kernels run on MI355X only: got a tensor on {inputs.device} (no CPU fallback)")
        if inputs.dim() != 4 or inputs.shape[1] != 1 or tuple(inputs.shape) != tuple(targets.shape):
            raise MisError(f"SegmentationLoss expects (N, 1, H, W) logits and targets of equal shape, got {tuple(inputs.shape)} / {tuple(targets.shape)}")
        x = inputs.contiguous().float()
        t = targets.to(x.device).contiguous().float()
        N, _, H, W = x.shape
        lib = load()
        ws = ops.workspace(lib.mis_segloss_workspace_bytes(N, H, W), x.device, "segloss")
        out = torch.empty(8, dtype=torch.float32, device=x.device)
        check(lib.mis_segloss_fwd(x.data_ptr(), t.data_ptr(), N, H, W, float(w_f1), float(w_ms), float(w_iou), ws.data_ptr(), out.data_ptr(),
                                  stream_ptr()), "mis_segloss_fwd")
        ctx.save_for_backward(t, out)
        ctx.ws = ws                      # the pyramid state of this forward (one loss evaluation in flight at a time)
        ctx.cfg = (N, H, W, inputs.shape, inputs.dtype, 1 if float(w_ms) != 0.0 else 0)
        return out[0].clone()

    @staticmethod
    def backward(ctx, g):
        t, out = ctx.saved_tensors
        N, H, W, shape, dtype, with_ms = ctx.cfg
        dx = torch.empty(N, 1, H, W, dtype=torch.float32, device=t.device)
        gg = g.contiguous().float().reshape(1)
        check(load().mis_segloss_bwd(t.data_ptr(), N, H, W, ctx.ws.data_ptr(), out.data_ptr(), gg.data_ptr(), dx.data_ptr(), with_ms, stream_ptr()),
              "mis_segloss_bwd")
        return dx.view(shape).to(dtype), None, None, None, None


class MSSSIMLoss(nn.Module):
    """1 - MS_SSIM(sigmoid(inputs), targets)   [loss.py:21-29]"""

    def forward(self, inputs, targets):
        return _SegLoss.apply(inputs, targets, 0.0, 1.0, 0.0)


class IoULoss(nn.Module):
    """1 - (sum(p*t) + eps) / (sum(p) + sum(t) - sum(p*t) + eps), p = sigmoid(inputs)   [loss.py:32-42]"""

    def __init__(self, epsilon=1e-7):
        super().__init__()
        if epsilon != 1e-7:
            raise NotImplementedError("IoULoss on MI355X: epsilon=1e-7 (the reference's value) is built")
        self.epsilon = epsilon

    def forward(self, inputs, targets):
        return _SegLoss.apply(inputs, targets, 0.0, 0.0, 1.0)


class F1Loss(nn.Module):
    """1 - 2*P*R / (P + R + eps), P = TP/(sum(p)+eps), R = TP/(sum(t)+eps)   [loss.py:45-56]"""

    def __init__(self, epsilon=1e-7):
        super().__init__()
        if epsilon != 1e-7:
            raise NotImplementedError("F1Loss on MI355X: epsilon=1e-7 (the reference's value) is built")
        self.epsilon = epsilon

    def forward(self, inputs, targets):
        return _SegLoss.apply(inputs, targets, 1.0, 0.0, 0.0)


class SegmentationLoss(nn.Module):
    """F1Loss + MSSSIMLoss + IoULoss in one fused forward / backward   [loss.py:58-70]"""

    def __init__(self):
        super().__init__()
        self.f1_loss = F1Loss()
        self.ms_ssim_loss = MSSSIMLoss()
        self.iou_loss = IoULoss()

    def forward(self, inputs, targets):
        return _SegLoss.apply(inputs, targets, 1.0, 1.0, 1.0)
