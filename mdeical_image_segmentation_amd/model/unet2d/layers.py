"""Mirror of the reference's 2-D building blocks (model/unet2d/layers.py:103-192) on the HIP kernels.

Each module keeps stock parameter containers (nn.Conv2d / nn.ConvTranspose2d: identical state-dict keys and
shapes) but its forward/backward run through libmisamd.  Inputs are (N, C, H, W) tensors of any layout; outputs are (N, C, H, W)-shaped
channels_last tensors in the compute dtype, i.e. NHWC memory, so that chained layers hand activations over without a layout pass.
The fused whole-network path (engine2d.UNet2DEngine, used by UNet / UNetModel) does not go through these
per-layer functions; they exist so that the blocks remain usable on their own.
CUDA tensors only - there is no CPU fallback."""
import os

import torch
from torch import nn

from ... import ops
from ..._lib import MisError

__all__ = ["DoubleConvolution", "DownSample", "UpSample", "CropAndConcat", "unetConv2", "unetUp", "unetUp_origin"]


def _compute_dtype():
    import os
    return torch.bfloat16 if os.environ.get("MISAMD_DTYPE", "f32").lower() in ("bf16", "bfloat16") else torch.float32


def _need_cuda(x):
    if x.device.type != "cuda":
        raise MisError("mdeical_image_segmentation_amd runs on MI355X only: got a tensor on %s (no CPU fallback)" % x.device)


def _to_nhwc(x, dtype, Cp=None):
    """(N, C, H, W) tensor -> contiguous (N, H, W, Cp) tensor of the compute dtype (Cp >= C: zero-padded channels).
    A channels_last tensor that already has the compute dtype - what every function below RETURNS - is used as it is (a free view):
    activations cross layer boundaries in NHWC memory, only the network input (NCHW fp32) is converted by a kernel."""
    N, C, H, W = x.shape
    Cp = C if Cp is None else Cp
    if Cp == C and x.dtype == dtype:
        v = x.permute(0, 2, 3, 1)
        if v.is_contiguous():
            return v
    y = torch.zeros(N, H, W, Cp, dtype=dtype, device=x.device) if Cp != C else torch.empty(N, H, W, Cp, dtype=dtype, device=x.device)
    if x.is_contiguous() and x.dtype == torch.float32:
        ops.nchw_to_nhwc(x, ops.View(y, 0, C))              # NCHW fp32 (the network input): layout kernel
    else:
        y[..., :C].copy_(x.permute(0, 2, 3, 1))             # any other stride pattern (e.g. a channel slice handed back by torch.cat's backward)
    return y


def _grad_nhwc(gy, dtype):
    """Incoming gradient (N, C, H, W) -> NHWC kernel operand.  torch.cat's backward hands every branch a channel slice (narrow) of one
    channels_last gradient: such a slice is passed to the kernels as a strided ops.View of its base (row stride = all channels), without a copy."""
    base = gy._base
    if base is not None and base.dim() == 4 and gy.dtype == dtype and base.dtype == dtype and gy.stride() == base.stride():
        N, C, H, W = gy.shape
        full = base.permute(0, 2, 3, 1)
        c0 = gy.storage_offset() - base.storage_offset()
        if full.is_contiguous() and tuple(full.shape[:3]) == (N, H, W) and 0 <= c0 and c0 + C <= full.shape[3] \
                and (c0 * gy.element_size()) % 16 == 0 and (full.shape[3] * gy.element_size()) % 16 == 0:
            return ops.View(full, c0, C)
    return _to_nhwc(gy, dtype)


def _to_nchw(y, C=None):
    """NHWC buffer -> (N, C, H, W)-shaped channels_last VIEW of its first C channels (no copy, same dtype)"""
    v = y.permute(0, 3, 1, 2)
    return v if C is None or C == y.shape[-1] else v[:, :C]


class _Conv3x3ReLU(torch.autograd.Function):
    """y = relu(conv2d(x, w, b, padding=1)) - reference layers.py:122-126."""

    @staticmethod
    def forward(ctx, x, w, b):
        _need_cuda(x)
        dt = _compute_dtype()
        N, Cin, H, W = x.shape
        Cout = w.shape[0]
        y = torch.empty(N, H, W, Cout, dtype=dt, device=x.device)
        if Cin <= 4:
            xin = x.contiguous(memory_format=torch.contiguous_format).float()
            ops.first_conv_fwd(xin, w.detach().contiguous(), b.detach(), y)
            wd = None
        else:
            xin = _to_nhwc(x, dt)
            wf = torch.empty(9, Cout, Cin, dtype=dt, device=x.device)
            wd = torch.empty(9, Cin, Cout, dtype=dt, device=x.device)
            ops.pack_conv_weight(w.detach().contiguous(), wf, wd)
            ops.conv_igemm(xin, wf, y, ksize=3, Cin=Cin, Cout=Cout, bias=b.detach(), relu=True)
        ctx.save_for_backward(xin, y)
        ctx.wd = wd
        ctx.shape = (N, Cin, H, W, Cout)
        return _to_nchw(y)

    @staticmethod
    def backward(ctx, gy):
        xin, y = ctx.saved_tensors
        N, Cin, H, W, Cout = ctx.shape
        dt = y.dtype
        # dL/d(pre-activation) = gy * (y > 0)   (never in place: gy may be a view of autograd's own buffer)
        g = torch.empty_like(y)
        ops.relu_mask(_to_nhwc(gy, dt), y, g)
        dw = torch.empty(Cout, Cin, 3, 3, dtype=torch.float32, device=gy.device)
        db = torch.empty(Cout, dtype=torch.float32, device=gy.device)
        if Cin <= 4:
            ops.first_conv_wgrad(xin, g, dw, db)
            return None, dw, db
        ops.wgrad(xin, g, dw, ksize=3, Cin=Cin, Cout=Cout)
        ops.colsum(g, db)
        dx = torch.empty(N, H, W, Cin, dtype=dt, device=gy.device)
        ops.conv_igemm(g, ctx.wd, dx, ksize=3, Cin=Cout, Cout=Cin)
        return _to_nchw(dx), dw, db


class _MaxPool2(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        _need_cuda(x)
        dt = _compute_dtype()
        N, C, H, W = x.shape
        xin = _to_nhwc(x, dt)
        y = torch.empty(N, H // 2, W // 2, C, dtype=dt, device=x.device)
        ops.maxpool2_fwd(xin, y)
        ctx.save_for_backward(xin)
        return _to_nchw(y)

    @staticmethod
    def backward(ctx, gy):
        (xin,) = ctx.saved_tensors
        g = _to_nhwc(gy, xin.dtype)
        dx = torch.empty_like(xin)
        ops.maxpool2_bwd(xin, g, dx, add=None, relu_mask=False)
        return _to_nchw(dx)


class _ConvT2x2(torch.autograd.Function):
    """y = conv_transpose2d(x, w, b, stride=2), w [Cin, Cout, 2, 2] - reference layers.py:165."""

    @staticmethod
    def forward(ctx, x, w, b):
        _need_cuda(x)
        dt = _compute_dtype()
        N, Cin, H, W = x.shape
        Cq = w.shape[1]
        xin = _to_nhwc(x, dt)
        wf = torch.empty(4 * Cq, Cin, dtype=dt, device=x.device)
        wd = torch.empty(Cin, 4 * Cq, dtype=dt, device=x.device)
        ops.pack_convt_weight(w.detach().contiguous(), wf, wd)
        y = torch.empty(N, 2 * H, 2 * W, Cq, dtype=dt, device=x.device)
        ops.conv_igemm(xin, wf, y, ksize=1, Cin=Cin, Cout=4 * Cq, bias=b.detach(), y0_mode=ops.OUT_SHUFFLE2)
        ctx.save_for_backward(xin)
        ctx.wd = wd
        ctx.shape = (N, Cin, H, W, Cq)
        return _to_nchw(y)

    @staticmethod
    def backward(ctx, gy):
        (xin,) = ctx.saved_tensors
        N, Cin, H, W, Cq = ctx.shape
        dt = xin.dtype
        # pixel-unshuffle the incoming gradient: (N, Cq, 2H, 2W) -> (N, H, W, 4*Cq) with column ab*Cq + c
        g = gy.contiguous(memory_format=torch.contiguous_format).float().view(N, Cq, H, 2, W, 2).permute(0, 2, 4, 3, 5, 1).reshape(N, H, W, 4 * Cq).to(dt).contiguous()
        dw = torch.empty(Cin, Cq, 2, 2, dtype=torch.float32, device=gy.device)
        db = torch.empty(Cq, dtype=torch.float32, device=gy.device)
        ops.wgrad(xin, g, dw, ksize=1, Cin=Cin, Cout=4 * Cq, dw_layout=1)
        ops.colsum(g, db, fold=4)
        dx = torch.empty(N, H, W, Cin, dtype=dt, device=gy.device)
        ops.conv_igemm(g, ctx.wd, dx, ksize=1, Cin=4 * Cq, Cout=Cin)
        return _to_nchw(dx), dw, db


class DoubleConvolution(nn.Module):
    """Conv3x3(p1,bias) -> ReLU -> Conv3x3(p1,bias) -> ReLU, no norm (reference layers.py:103-133)."""

    def __init__(self, in_channels: int, out_channels: int):
        super().__init__()
        self.first = nn.Conv2d(in_channels, out_channels, kernel_size=3, padding=1)
        self.act1 = nn.ReLU()
        self.second = nn.Conv2d(out_channels, out_channels, kernel_size=3, padding=1)
        self.act2 = nn.ReLU()

    def forward(self, x: torch.Tensor):
        x = _Conv3x3ReLU.apply(x, self.first.weight, self.first.bias)
        return _Conv3x3ReLU.apply(x, self.second.weight, self.second.bias)


class DownSample(nn.Module):
    """MaxPool2d(2) (reference layers.py:136-150)."""

    def __init__(self):
        super().__init__()
        self.pool = nn.MaxPool2d(2)

    def forward(self, x: torch.Tensor):
        return _MaxPool2.apply(x)


class UpSample(nn.Module):
    """ConvTranspose2d(k2, s2) (reference layers.py:153-168)."""

    def __init__(self, in_channels: int, out_channels: int):
        super().__init__()
        self.up = nn.ConvTranspose2d(in_channels, out_channels, kernel_size=2, stride=2)

    def forward(self, x: torch.Tensor):
        return _ConvT2x2.apply(x, self.up.weight, self.up.bias)


class CropAndConcat(nn.Module):
    """center-crop the skip to x's size, cat([x, skip], 1): up-sampled channels first (reference layers.py:171-192).
    Pure indexing (tensor plumbing); inside the fused engine it does not exist at all."""

    def forward(self, x: torch.Tensor, contracting_x: torch.Tensor):
        h, w = x.shape[2], x.shape[3]
        H, W = contracting_x.shape[2], contracting_x.shape[3]
        top = int(round((H - h) / 2.0))
        left = int(round((W - w) / 2.0))
        return torch.cat([x, contracting_x[:, :, top:top + h, left:left + w]], dim=1)


def _out_of_scope(name, where):
    class _Stub(nn.Module):
        def __init__(self, *a, **k):
            raise NotImplementedError(f"{name} ({where}) is outside the accelerated hot path (SURVEY.md §8f): "
                                      "only the classic UNet is built")
    _Stub.__name__ = name
    return _Stub


def _pad64(c):
    return (c + 63) // 64 * 64


def _pad_chunk(c, dt):
    e = 8 if dt == torch.bfloat16 else 4
    return (c + e - 1) // e * e


def _conv3x3_cols(x, w_tap_out_in, y, n_out, bias=None, **kw):
    """3x3 convolution with the [tap][out][in] operand `w_tap_out_in`; an output width that is not a multiple of 128 (UNet 3+'s 320) is cut into
    a 128-multiple part for the 256 px x 128 ch MFMA tiles (≈1.05 PFLOP/s) and a remainder for the 64-column tiles (≈0.86), instead of running all
    of it on the narrow tiles.  MISAMD_NO_COL_SPLIT=1 disables the cut."""
    cin = w_tap_out_in.shape[2]
    c0 = (n_out // 128) * 128
    if c0 == 0 or c0 == n_out or os.environ.get("MISAMD_NO_COL_SPLIT") == "1":
        ops.conv_igemm(x, w_tap_out_in, y, ksize=3, Cin=cin, Cout=n_out, bias=bias, **kw)
        return
    for lo, hi in ((0, c0), (c0, n_out)):
        ops.conv_igemm(x, w_tap_out_in[:, lo:hi].contiguous(), ops.View(y, lo, hi - lo), ksize=3, Cin=cin, Cout=hi - lo,
                       bias=None if bias is None else bias[lo:hi].contiguous(), **kw)


def _bn_relu_out(z, scale, shift, out):
    """y = relu(z * scale + shift): into a fresh buffer, or straight into a channel slice (ops.View) of a concatenation buffer, which makes the
    reference's torch.cat of the five decoder branches (unet.py:237-243 etc.) a no-op; returns the (N, C, H, W) channels_last view of the result"""
    if out is None:
        y = torch.empty_like(z)
        ops.affine_act(z, y, scale, shift, relu=True)
        return _to_nchw(y)
    ops.affine_act(z, out, scale, shift, relu=True)
    return out.t[..., out.c0:out.c0 + out.C].permute(0, 3, 1, 2)


class _CatSlices(torch.autograd.Function):
    """torch.cat(parts, 1) when every part already IS its channel slice of `buf` (written there by _bn_relu_out): forward returns the buffer, backward
    hands each branch its slice of the gradient as a view (read in place by _grad_nhwc)"""

    @staticmethod
    def forward(ctx, buf, *parts):
        ctx.widths = [p.shape[1] for p in parts]
        return buf.view_as(buf)

    @staticmethod
    def backward(ctx, g):
        outs, c = [], 0
        for wdt in ctx.widths:
            outs.append(g[:, c:c + wdt])
            c += wdt
        return (None, *outs)


class _Conv3x3BNReLU(torch.autograd.Function):
    """y = relu(batch_norm(conv2d(x, w, b, padding=1))) - reference layers.py:17-25 (one `conv%d` Sequential of unetConv2).
    Input channels are zero-padded to a multiple of 64 (the K tile of the MFMA kernels); running statistics are updated in place."""

    @staticmethod
    def forward(ctx, x, w, b, gamma, beta, running_mean, running_var, training, eps, momentum, out=None):
        _need_cuda(x)
        dt = _compute_dtype()
        dev = x.device
        N, Cin, H, W = x.shape
        Cout, Cp = w.shape[0], _pad64(Cin)
        if Cout % 64:
            raise MisError(f"unetConv2: out_size {Cout} must be a multiple of 64")
        xin = _to_nhwc(x, dt, Cp)
        wpad = w.detach().float()
        if Cp != Cin:
            wpad = torch.zeros(Cout, Cp, 3, 3, dtype=torch.float32, device=dev)
            wpad[:, :Cin] = w.detach()
        wf = torch.empty(9, Cout, Cp, dtype=dt, device=dev)
        wd = torch.empty(9, Cp, Cout, dtype=dt, device=dev)
        ops.pack_conv_weight(wpad.contiguous(), wf, wd)
        z = torch.empty(N, H, W, Cout, dtype=dt, device=dev)
        _conv3x3_cols(xin, wf, z, Cout, bias=b.detach().float())
        f32 = dict(dtype=torch.float32, device=dev)
        scale, shift = torch.empty(N, Cout, **f32), torch.empty(N, Cout, **f32)
        mean, rstd = torch.empty(Cout, **f32), torch.empty(Cout, **f32)
        s = sq = None
        if training:
            s, sq = torch.empty(N, Cout, **f32), torch.empty(N, Cout, **f32)
            ops.chanstats(z, s, sq)
        ops.bn_fwd_finalize(s, sq, N, Cout, N * H * W, gamma.detach().float(), beta.detach().float(), running_mean, running_var, training,
                            scale, shift, mean, rstd, eps=eps, momentum=momentum)
        ctx.save_for_backward(xin, z, scale, shift, mean, rstd, gamma.detach().float())
        ctx.wd, ctx.training = wd, training
        ctx.shape = (N, Cin, H, W, Cout, Cp)
        return _bn_relu_out(z, scale, shift, out)

    @staticmethod
    def backward(ctx, gy):
        xin, z, scale, shift, mean, rstd, gamma = ctx.saved_tensors
        N, Cin, H, W, Cout, Cp = ctx.shape
        dt, dev = z.dtype, gy.device
        f32 = dict(dtype=torch.float32, device=dev)
        g = _grad_nhwc(gy, dt)
        S1, S2 = torch.empty(N, Cout, **f32), torch.empty(N, Cout, **f32)
        ops.bn_bwd_stats(g, z, scale, shift, S1, S2)            # the ReLU mask (y > 0) is recomputed from z: no masked copy of the gradient
        p, q, r = (torch.empty(N, Cout, **f32) for _ in range(3))
        dgamma, dbeta = torch.empty(Cout, **f32), torch.empty(Cout, **f32)
        ops.bn_bwd_finalize(S1, S2, mean, rstd, gamma, N, Cout, N * H * W, ctx.training, p, q, r, dgamma, dbeta)
        dz = torch.empty_like(z)
        ops.bn_bwd_apply(g, z, scale, shift, p, q, r, dz)
        dwp = torch.empty(Cout, Cp, 3, 3, **f32)
        db = torch.empty(Cout, **f32)
        ops.wgrad(xin, dz, dwp, ksize=3, Cin=Cp, Cout=Cout, dbias=db)
        dw = dwp[:, :Cin].contiguous() if Cp != Cin else dwp
        dx = None
        if ctx.needs_input_grad[0]:
            dxp = torch.empty(N, H, W, Cp, dtype=dt, device=dev)
            _conv3x3_cols(dz, ctx.wd, dxp, Cp)
            dx = _to_nchw(dxp, Cin)
        return dx, dw, db, dgamma, dbeta, None, None, None, None, None, None


class _MaxPoolCeil(torch.autograd.Function):
    """nn.MaxPool2d(k, k, ceil_mode=True) - reference unet.py:170-171 etc. (UNet 3+ encoder-to-decoder skips)"""

    @staticmethod
    def forward(ctx, x, k):
        _need_cuda(x)
        dt = _compute_dtype()
        N, C, H, W = x.shape
        xin = _to_nhwc(x, dt)
        y = torch.empty(N, (H + k - 1) // k, (W + k - 1) // k, C, dtype=dt, device=x.device)
        ops.maxpoolk_fwd(xin, y, k)
        ctx.save_for_backward(xin)
        ctx.k = k
        return _to_nchw(y)

    @staticmethod
    def backward(ctx, gy):
        (xin,) = ctx.saved_tensors
        dx = torch.empty_like(xin)
        ops.maxpoolk_bwd(xin, _to_nhwc(gy, xin.dtype), dx, ctx.k)
        return _to_nchw(dx), None


class _BilinearUp(torch.autograd.Function):
    """nn.Upsample(scale_factor=s, mode='bilinear') (align_corners=False) - reference unet.py:190 etc."""

    @staticmethod
    def forward(ctx, x, s):
        _need_cuda(x)
        dt = _compute_dtype()
        N, C, H, W = x.shape
        Cp = _pad_chunk(C, dt)                  # the kernels move 16-byte channel chunks: few-channel maps (deep-supervision logits) are padded
        y = torch.empty(N, H * s, W * s, Cp, dtype=dt, device=x.device)
        ops.bilinear_up_fwd(_to_nhwc(x, dt, Cp), y, s)
        ctx.cfg = (N, C, Cp, H, W, s, dt)
        return _to_nchw(y, C)

    @staticmethod
    def backward(ctx, gy):
        N, C, Cp, H, W, s, dt = ctx.cfg
        dxp = torch.empty(N, H, W, Cp, dtype=dt, device=gy.device)
        ops.bilinear_up_bwd(_to_nhwc(gy, dt, Cp), dxp, s)
        return _to_nchw(dxp, C), None


class _UpConv3x3BNReLU(torch.autograd.Function):
    """y = relu(batch_norm(conv2d(upsample_bilinear_s(x), w, b, padding=1))) - the decoder-to-decoder branches of UNet 3+ (reference unet.py:190-192 etc.)
    WITHOUT the upsampled tensor: the channel contraction commutes with the interpolation, so it runs at the low resolution as a 1x1 GEMM with
    9*Cout columns (s^2 x fewer FLOPs than the reference's order) and mis_upconv_gather_fwd/bwd interpolate the nine tap products (csrc/upconv.hip).
    Same arithmetic up to fp32 summation order."""

    @staticmethod
    def forward(ctx, x, s, w, b, gamma, beta, running_mean, running_var, training, eps, momentum, out=None):
        _need_cuda(x)
        dt = _compute_dtype()
        dev = x.device
        N, Cin, h, wl = x.shape
        Cout, Cp = w.shape[0], _pad64(Cin)
        if Cout % 64:
            raise MisError(f"up-branch conv: out channels {Cout} must be a multiple of 64")
        H, W = h * s, wl * s
        xin = _to_nhwc(x, dt, Cp)
        wpad = w.detach().float()
        if Cp != Cin:
            wpad = torch.zeros(Cout, Cp, 3, 3, dtype=torch.float32, device=dev)
            wpad[:, :Cin] = w.detach()
        wf = torch.empty(9, Cout, Cp, dtype=dt, device=dev)             # [tap][co][ci] = the 1x1 filter bank of the 9*Cout-column GEMM
        ops.pack_conv_weight(wpad.contiguous(), wf)
        zt = torch.empty(N, h, wl, 9 * Cout, dtype=dt, device=dev)
        ops.conv_igemm(xin, wf, zt, ksize=1, Cin=Cp, Cout=9 * Cout)
        z = torch.empty(N, H, W, Cout, dtype=dt, device=dev)
        ops.upconv_gather_fwd(zt, z, s, Cout, bias=b.detach().float())
        del zt
        f32 = dict(dtype=torch.float32, device=dev)
        scale, shift = torch.empty(N, Cout, **f32), torch.empty(N, Cout, **f32)
        mean, rstd = torch.empty(Cout, **f32), torch.empty(Cout, **f32)
        sm = sq = None
        if training:
            sm, sq = torch.empty(N, Cout, **f32), torch.empty(N, Cout, **f32)
            ops.chanstats(z, sm, sq)
        ops.bn_fwd_finalize(sm, sq, N, Cout, N * H * W, gamma.detach().float(), beta.detach().float(), running_mean, running_var, training,
                            scale, shift, mean, rstd, eps=eps, momentum=momentum)
        ctx.save_for_backward(xin, z, scale, shift, mean, rstd, gamma.detach().float(), wpad)
        ctx.training = training
        ctx.shape = (N, Cin, h, wl, Cout, Cp, s)
        return _bn_relu_out(z, scale, shift, out)

    @staticmethod
    def backward(ctx, gy):
        xin, z, scale, shift, mean, rstd, gamma, wpad = ctx.saved_tensors
        N, Cin, h, wl, Cout, Cp, s = ctx.shape
        H, W = h * s, wl * s
        dt, dev = z.dtype, gy.device
        f32 = dict(dtype=torch.float32, device=dev)
        g = _grad_nhwc(gy, dt)
        S1, S2 = torch.empty(N, Cout, **f32), torch.empty(N, Cout, **f32)
        ops.bn_bwd_stats(g, z, scale, shift, S1, S2)
        p, q, r = (torch.empty(N, Cout, **f32) for _ in range(3))
        dgamma, dbeta = torch.empty(Cout, **f32), torch.empty(Cout, **f32)
        ops.bn_bwd_finalize(S1, S2, mean, rstd, gamma, N, Cout, N * H * W, ctx.training, p, q, r, dgamma, dbeta)
        dz = torch.empty_like(z)
        ops.bn_bwd_apply(g, z, scale, shift, p, q, r, dz)
        del g
        db = torch.empty(Cout, **f32)
        ops.colsum(dz, db)
        dzt = torch.empty(N, h, wl, 9 * Cout, dtype=dt, device=dev)
        ops.upconv_gather_bwd(dz, dzt, s, Cout)
        dwk = torch.empty(9 * Cout, Cp, 1, 1, **f32)
        ops.wgrad(xin, dzt, dwk, ksize=1, Cin=Cp, Cout=9 * Cout)
        dw = dwk.view(3, 3, Cout, Cp).permute(2, 3, 0, 1)[:, :Cin].contiguous()
        dx = None
        if ctx.needs_input_grad[0]:
            wd = torch.empty(Cp, 3, 3, Cout, dtype=dt, device=dev)                         # [ci][tap*Cout + co]: filter bank of the transposed GEMM
            wd.copy_(wpad.permute(1, 2, 3, 0))                                             # one permuting + casting copy
            dxp = torch.empty(N, h, wl, Cp, dtype=dt, device=dev)
            ops.conv_igemm(dzt, wd, dxp, ksize=1, Cin=9 * Cout, Cout=Cp)
            dx = _to_nchw(dxp, Cin)
        return dx, None, dw, db, dgamma, dbeta, None, None, None, None, None, None


def up_conv_bn_relu(x, s, conv, bn, module_training, out=None):
    """relu(bn(conv3x3(upsample_bilinear(x, s)))) - fused form (default) or, with MISAMD_UPCONV_UNFUSED=1, the reference's literal order"""
    if os.environ.get("MISAMD_UPCONV_UNFUSED") == "1":
        return conv_bn_relu(_BilinearUp.apply(x, s), conv, bn, module_training, out)
    training = module_training or not bn.track_running_stats
    if module_training and bn.track_running_stats:
        bn.num_batches_tracked += 1
    return _UpConv3x3BNReLU.apply(x, s, conv.weight, conv.bias, bn.weight, bn.bias, bn.running_mean, bn.running_var, training, bn.eps, bn.momentum, out)


class _Conv3x3Plain(torch.autograd.Function):
    """y = conv2d(x, w, b, padding=1) with ANY number of output channels (the 3x3 output head of UNet 3+, unet.py:325): the filter bank is
    zero-padded to 64 output channels for the MFMA kernels and the result sliced."""

    @staticmethod
    def forward(ctx, x, w, b):
        _need_cuda(x)
        dt = _compute_dtype()
        dev = x.device
        N, Cin, H, W = x.shape
        Cout, Cp, Op = w.shape[0], _pad64(Cin), _pad64(w.shape[0])
        xin = _to_nhwc(x, dt, Cp)
        wpad = torch.zeros(Op, Cp, 3, 3, dtype=torch.float32, device=dev)
        wpad[:Cout, :Cin] = w.detach()
        bpad = torch.zeros(Op, dtype=torch.float32, device=dev)
        bpad[:Cout] = b.detach()
        wf = torch.empty(9, Op, Cp, dtype=dt, device=dev)
        wd = torch.empty(9, Cp, Op, dtype=dt, device=dev)
        ops.pack_conv_weight(wpad, wf, wd)
        y = torch.empty(N, H, W, Op, dtype=dt, device=dev)
        ops.conv_igemm(xin, wf, y, ksize=3, Cin=Cp, Cout=Op, bias=bpad)
        ctx.save_for_backward(xin)
        ctx.wd = wd
        ctx.shape = (N, Cin, H, W, Cout, Cp, Op)
        return _to_nchw(y, Cout).float().contiguous()                # the logits leave as fp32 NCHW (tiny: Cout channels of the padded result)

    @staticmethod
    def backward(ctx, gy):
        (xin,) = ctx.saved_tensors
        N, Cin, H, W, Cout, Cp, Op = ctx.shape
        dt, dev = xin.dtype, gy.device
        g = _to_nhwc(gy, dt, Op)
        dwp = torch.empty(Op, Cp, 3, 3, dtype=torch.float32, device=dev)
        dbp = torch.empty(Op, dtype=torch.float32, device=dev)
        ops.wgrad(xin, g, dwp, ksize=3, Cin=Cp, Cout=Op, dbias=dbp)
        dxp = torch.empty(N, H, W, Cp, dtype=dt, device=dev)
        _conv3x3_cols(g, ctx.wd, dxp, Cp)
        return _to_nchw(dxp, Cin), dwp[:Cout, :Cin].contiguous(), dbp[:Cout].contiguous()


class unetConv2(nn.Module):
    """n x [Conv2d(ks 3, s 1, p 1, bias) -> BatchNorm2d -> ReLU] (or without the norm), kaiming-normal init - the reference's
    "BN" double-conv block (model/unet2d/layers.py:8-46).  Containers are stock nn.Sequential(Conv2d, BatchNorm2d, ReLU) named
    conv1..convN, so state-dict keys (incl. running_mean / running_var / num_batches_tracked) equal the reference's."""

    def __init__(self, in_size, out_size, is_batchnorm, n=2, ks=3, stride=1, padding=1):
        super().__init__()
        if ks != 3 or stride != 1 or padding != 1:
            raise NotImplementedError("unetConv2 on MI355X: only ks=3, stride=1, padding=1 (every use in the reference) is built")
        from .init_weights import init_weights
        self.n, self.ks, self.stride, self.padding = n, ks, stride, padding
        self.is_batchnorm = bool(is_batchnorm)
        for i in range(1, n + 1):
            mods = [nn.Conv2d(in_size, out_size, ks, stride, padding)]
            if is_batchnorm:
                mods.append(nn.BatchNorm2d(out_size))
            mods.append(nn.ReLU(inplace=True))
            setattr(self, "conv%d" % i, nn.Sequential(*mods))
            in_size = out_size
        for m in self.children():
            init_weights(m, init_type="kaiming")

    def forward(self, inputs):
        x = inputs
        for i in range(1, self.n + 1):
            seq = getattr(self, "conv%d" % i)
            conv = seq[0]
            if not self.is_batchnorm:
                x = _Conv3x3ReLU.apply(x, conv.weight, conv.bias)
                continue
            x = conv_bn_relu(x, conv, seq[1], self.training)
        return x


def conv_bn_relu(x, conv, bn, module_training, out=None):
    """relu(bn(conv3x3(x))) through the HIP path with nn.BatchNorm2d's train / eval semantics (running statistics, num_batches_tracked)"""
    training = module_training or not bn.track_running_stats
    if module_training and bn.track_running_stats:
        bn.num_batches_tracked += 1
    return _Conv3x3BNReLU.apply(x, conv.weight, conv.bias, bn.weight, bn.bias, bn.running_mean, bn.running_var, training, bn.eps, bn.momentum, out)


unetUp = _out_of_scope("unetUp", "model/unet2d/layers.py:49-74")
unetUp_origin = _out_of_scope("unetUp_origin", "model/unet2d/layers.py:76-101")
