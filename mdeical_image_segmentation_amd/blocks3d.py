"""Stand-alone 3-D building blocks on MI355X: the arithmetic behind `model.unet3d.buildingblocks.{SingleConv, DoubleConv, ResNetBlock, Encoder, Decoder}`
when a block is called ON ITS OWN, and behind the 3-D U-Nets whenever the fused engines (engine3d*.py) do not cover the configuration (channel counts that
are not multiples of 64, in_channels > 1, AvgPool3d / other pooling windows, grids that are not divisible by 2^(levels-1), layer orders other than 'gcr').

Reference behaviour: model/unet3d/buildingblocks.py:14-113 (`create_conv` order strings), :116-159 SingleConv, :162-252 DoubleConv, :255-325 ResNetBlock,
:365-439 Encoder, :442-550 Decoder, :553-673 upsampling (`F.interpolate(size=encoder_features.size()[2:])`).

Layout: every activation lives channels-last, (N, D, H, W, Cp) with Cp = C rounded up to 64 and the padding kept exactly 0 (zero weights, zero scale/shift),
so every convolution runs on the MFMA implicit-GEMM kernels (ops.conv_igemm / ops.wgrad) whatever C is.  Each function below is a torch.autograd.Function
whose forward AND backward are HIP kernels of libmisamd; torch is the allocator and the tape, nothing else.  There is no CPU path.
"""
import os

import torch

from . import ops
from ._lib import MisError
from .ops import View

ACT_CODES = {"r": (ops.ACT_RELU, 0.0), "l": (ops.ACT_LEAKY, 0.01), "e": (ops.ACT_ELU, 1.0)}


def compute_dtype(name=None):
    name = (name or os.environ.get("MISAMD_DTYPE", "f32")).lower()
    if name in ("bf16", "bfloat16"):
        return torch.bfloat16
    if name in ("f32", "fp32", "float32"):
        return torch.float32
    raise MisError(f"compute dtype must be 'f32' or 'bf16', got {name!r}")


def pad64(c):
    return (c + 63) // 64 * 64


class CL:
    """a channels-last activation: t = (N, D, H, W, pad64(C)) tensor on the tape, C = its logical channel count"""

    def __init__(self, t, C):
        self.t, self.C = t, C

    @property
    def grid(self):
        return tuple(self.t.shape[:4])


def _need_cuda(x):
    if x.device.type != "cuda":
        raise MisError(f"the 3-D building blocks run on MI355X only: got a tensor on {x.device} (no CPU fallback)")
    if x.dim() not in (4, 5):
        raise MisError(f"expected a (N, C, D, H, W) tensor (or (N, C, H, W) for the 2-D variants of the blocks), got {tuple(x.shape)}")


# ---- layout ------------------------------------------------------------------------------------------------------------------------------
class _ToCL(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, dtype):
        x = x.contiguous().float()
        N, C = x.shape[:2]
        y = torch.zeros(N, *x.shape[2:], pad64(C), dtype=dtype, device=x.device)
        ops.nchw_to_nhwc(x, View(y, 0, C))
        ctx.C = C
        return y

    @staticmethod
    def backward(ctx, g):
        g = g.contiguous()
        N, D, H, W, _ = g.shape
        dx = torch.empty(N, ctx.C, D, H, W, dtype=torch.float32, device=g.device)
        ops.nhwc_to_nchw(View(g, 0, ctx.C), dx)
        return dx, None


class _FromCL(torch.autograd.Function):
    @staticmethod
    def forward(ctx, t, C):
        t = t.contiguous()
        N, D, H, W, _ = t.shape
        y = torch.empty(N, C, D, H, W, dtype=torch.float32, device=t.device)
        ops.nhwc_to_nchw(View(t, 0, C), y)
        ctx.C, ctx.dtype, ctx.Cp = C, t.dtype, t.shape[-1]
        return y

    @staticmethod
    def backward(ctx, g):
        g = g.contiguous().float()
        N, _, D, H, W = g.shape
        dt = torch.zeros(N, D, H, W, ctx.Cp, dtype=ctx.dtype, device=g.device)
        ops.nchw_to_nhwc(g, View(dt, 0, ctx.C))
        return dt, None


def to_cl(x, dtype=None):
    """(N, C, D, H, W) -> channels-last activation; a 2-D tensor (N, C, H, W) (UNet2D / ResidualUNet2D, is3d=False blocks) travels as a volume of depth 1"""
    _need_cuda(x)
    if x.dim() == 4:
        x = x.unsqueeze(2)
    return CL(_ToCL.apply(x, compute_dtype() if dtype is None else dtype), x.shape[1])


def from_cl(a, two_d=False):
    y = _FromCL.apply(a.t, a.C)
    return y.squeeze(2) if two_d else y


# ---- GroupNorm statistics shared by the two normalising functions ----------------------------------------------------------------------------
def _gn_forward(t, C, G, gamma, beta):
    N, Cp = t.shape[0], t.shape[-1]
    dev = t.device
    s, sq = torch.empty(N, Cp, device=dev), torch.empty(N, Cp, device=dev)
    ops.chanstats(t, s, sq)
    scale, shift = torch.empty(N, Cp, device=dev), torch.empty(N, Cp, device=dev)
    mean, rstd = torch.empty(N, G, device=dev), torch.empty(N, G, device=dev)
    ops.gn_fwd_finalize_ld(s, sq, N, C, Cp, G, t[0, ..., 0].numel(), gamma, beta, scale, shift, mean, rstd)
    return scale, shift, mean, rstd


def _gn_backward(dz, x, C, G, gamma, mean, rstd):
    """dz = dL/d(GroupNorm output), x = its input -> (dx, dgamma, dbeta)"""
    N, Cp = x.shape[0], x.shape[-1]
    dev = x.device
    grid = tuple(x.shape[:4])
    S1, S2 = torch.empty(N, Cp, device=dev), torch.empty(N, Cp, device=dev)
    ops.gn_bwd_stats(dz, x, Cp, False, grid, S1, S2, Cp, 0)
    p, q, r = (torch.empty(N, Cp, device=dev) for _ in range(3))
    dgamma, dbeta = torch.empty(C, device=dev), torch.empty(C, device=dev)
    ops.gn_bwd_finalize_ld(S1, S2, mean, rstd, gamma, N, C, Cp, G, x[0, ..., 0].numel(), p, q, r, dgamma, dbeta)
    dx = torch.empty_like(x)
    ops.gn_bwd_apply(dz, x, Cp, False, grid, p, q, r, Cp, 0, dx)
    return dx, dgamma, dbeta


# ---- BatchNorm3d ('b' of create_conv, buildingblocks.py:93-104): a per-channel affine from batch (training) or running (eval) statistics -------------------------
def _padded(v, Cp, fill=0.0):
    out = torch.full((Cp,), fill, dtype=torch.float32, device=v.device)
    out[:v.numel()] = v.detach().float()
    return out


def _bn_forward(t, C, bn):
    """scale / shift tables [N][Cp] (the same per-channel values for every sample, so that the (n, c)-affine consumers are shared with GroupNorm), batch mean / rstd [Cp];
    training mode updates the module's running statistics (momentum, unbiased variance) and num_batches_tracked exactly like nn.BatchNorm3d"""
    N, Cp = t.shape[0], t.shape[-1]
    dev = t.device
    if bn.num_features != C or not bn.affine:
        raise MisError(f"BatchNorm3d(num_features = {C}, affine) expected, got {bn}")
    training = bn.training or not bn.track_running_stats
    count = float(N * t[0, ..., 0].numel())
    s = sq = None
    if training:
        s, sq = torch.empty(N, Cp, device=dev), torch.empty(N, Cp, device=dev)
        ops.chanstats(t, s, sq)
    rm = rv = None
    if bn.track_running_stats:
        rm, rv = _padded(bn.running_mean, Cp), _padded(bn.running_var, Cp, 1.0)
    momentum = bn.momentum
    if training and bn.track_running_stats:
        bn.num_batches_tracked += 1
        if momentum is None:                       # cumulative moving average (nn.BatchNorm semantics)
            momentum = 1.0 / float(bn.num_batches_tracked)
    scale, shift = torch.empty(N, Cp, device=dev), torch.empty(N, Cp, device=dev)
    mean, rstd = torch.empty(Cp, device=dev), torch.empty(Cp, device=dev)
    ops.bn_fwd_finalize(s, sq, N, Cp, count, _padded(bn.weight, Cp), _padded(bn.bias, Cp), rm, rv, training, scale, shift, mean, rstd, eps=bn.eps,
                        momentum=0.0 if momentum is None else momentum)
    if training and bn.track_running_stats:
        bn.running_mean.copy_(rm[:C])
        bn.running_var.copy_(rv[:C])
    return scale, shift, mean, rstd, training, count


def _bn_backward(dz, x, C, gamma, mean, rstd, training, count):
    """dz = dL/d(BatchNorm output), x = its input -> (dx, dgamma, dbeta)"""
    N, Cp = x.shape[0], x.shape[-1]
    dev = x.device
    grid = tuple(x.shape[:4])
    S1, S2 = torch.empty(N, Cp, device=dev), torch.empty(N, Cp, device=dev)
    ops.gn_bwd_stats(dz, x, Cp, False, grid, S1, S2, Cp, 0)
    p, q, r = (torch.empty(N, Cp, device=dev) for _ in range(3))
    dgamma, dbeta = torch.empty(Cp, device=dev), torch.empty(Cp, device=dev)
    ops.bn_bwd_finalize(S1, S2, mean, rstd, _padded(gamma, Cp), N, Cp, count, training, p, q, r, dgamma, dbeta)
    dx = torch.empty_like(x)
    ops.gn_bwd_apply(dz, x, Cp, False, grid, p, q, r, Cp, 0, dx)
    return dx, dgamma[:C].contiguous(), dbeta[:C].contiguous()


# ---- [GroupNorm | BatchNorm ->] Conv3d [+ bias] [-> ReLU] -------------------------------------------------------------------------------------------
class _Conv(torch.autograd.Function):
    """x (N, D, H, W, Cinp) -> y (N, D, H, W, Coutp).  weight: the reference's (Cout, Cin, k, k, k) fp32 parameter, k = 3 (padding 1) or 1.  A GroupNorm in
    front of the convolution is folded into the operand staging of the forward and of the weight-gradient kernel (the normalised tensor is never written);
    a ReLU behind it is the forward's epilogue."""

    @staticmethod
    def forward(ctx, x, weight, bias, gamma, beta, cin, cout, groups, relu, bn=None):
        x = x.contiguous()
        dev, dt = x.device, x.dtype
        ks = weight.shape[-1]
        nd = weight.dim() - 2                                  # 3: Conv3d; 2: Conv2d (is3d=False blocks) on a depth-1 volume, run by the 2-D kernels on (N, H, W, C) views
        cinp, coutp = x.shape[-1], pad64(cout)
        if nd not in (2, 3) or tuple(weight.shape) != (cout, cin) + (ks,) * nd or ks not in (1, 3) or cinp != pad64(cin):
            raise MisError(f"conv block: weight {tuple(weight.shape)} does not fit {cin} -> {cout} channels (kernel 1 or 3)")
        if nd == 2 and x.shape[1] != 1:
            raise MisError(f"conv block: a Conv2d weight needs a depth-1 activation, got grid {tuple(x.shape[1:4])}")
        wpad = torch.zeros(coutp, cinp, *((ks,) * nd), device=dev)
        wpad[:cout, :cin] = weight.detach()
        taps = ks ** nd
        wf = torch.empty(taps, coutp, cinp, dtype=dt, device=dev)
        wd = torch.empty(taps, cinp, coutp, dtype=dt, device=dev)
        ops.pack_conv_weight(wpad, wf, wd)
        scale = shift = mean = rstd = None
        ctx.bn = None
        if bn is not None:
            scale, shift, mean, rstd, training, count = _bn_forward(x, cin, bn)
            ctx.bn = (training, count)
        elif gamma is not None:
            scale, shift, mean, rstd = _gn_forward(x, cin, groups, gamma.detach().float(), beta.detach().float())
        bpad = None
        if bias is not None:
            bpad = torch.zeros(coutp, device=dev)
            bpad[:cout] = bias.detach()
        y = torch.empty(*x.shape[:4], coutp, dtype=dt, device=dev)
        v = (lambda t: t) if nd == 3 else (lambda t: t.view(t.shape[0], *t.shape[2:]))      # depth-1 volume -> image
        ops.conv_igemm(v(x), wf, v(y), ksize=ks, Cin=cinp, Cout=coutp, bias=bpad, relu=relu, in_scale=scale, in_shift=shift)
        ctx.save_for_backward(x, y if relu else None, wd, scale, shift, mean, rstd, None if gamma is None else gamma.detach().float())
        ctx.cfg = (cin, cout, groups, relu, ks, bias is not None, nd)
        return y

    @staticmethod
    def backward(ctx, g):
        x, y, wd, scale, shift, mean, rstd, gamma = ctx.saved_tensors
        cin, cout, groups, relu, ks, has_bias, nd = ctx.cfg
        dev = x.device
        cinp, coutp = x.shape[-1], g.shape[-1]
        g = g.contiguous()
        if relu:
            gm = torch.empty_like(g)
            ops.relu_mask(g, y, gm)
            g = gm
        v = (lambda t: t) if nd == 3 else (lambda t: t.view(t.shape[0], *t.shape[2:]))
        dwp = torch.empty(coutp, cinp, *((ks,) * nd), device=dev)
        dbp = torch.empty(coutp, device=dev) if has_bias else None
        ops.wgrad(v(x), v(g), dwp, ksize=ks, Cin=cinp, Cout=coutp, in_scale=scale, in_shift=shift, dbias=dbp)
        dw = dwp[:cout, :cin].contiguous()
        db = dbp[:cout].contiguous() if has_bias else None
        dx = dgamma = dbeta = None
        if ctx.needs_input_grad[0] or gamma is not None:
            dn = torch.empty_like(x)
            ops.conv_igemm(v(g), wd, v(dn), ksize=ks, Cin=coutp, Cout=cinp)
            if ctx.bn is not None:
                dx, dgamma, dbeta = _bn_backward(dn, x, cin, gamma, mean, rstd, *ctx.bn)
            elif gamma is not None:
                dx, dgamma, dbeta = _gn_backward(dn, x, cin, groups, gamma, mean, rstd)
            else:
                dx = dn
        return dx, dw, db, dgamma, dbeta, None, None, None, None, None


def conv(a, weight, bias=None, gn=None, relu=False, bn=None):
    """gn / bn = the nn.GroupNorm / nn.BatchNorm3d in front of the convolution, or None: folded into the operand staging"""
    cout, cin = weight.shape[:2]
    if cin != a.C:
        raise MisError(f"conv block: input has {a.C} channels, the weight expects {cin}")
    gamma = beta = None
    groups = 1
    if gn is not None:
        gamma, beta, groups = gn.weight, gn.bias, gn.num_groups
        if gn.num_channels != cin or abs(gn.eps - 1e-5) > 0:
            raise MisError("conv block: GroupNorm(num_channels = conv in_channels, eps = 1e-5) expected")
    if bn is not None:
        gamma, beta = bn.weight, bn.bias
    return CL(_Conv.apply(a.t, weight, bias, gamma, beta, cin, cout, groups, relu, bn), cout)


# ---- [GroupNorm ->] activation behind a convolution ('cge', 'cl', 'crg', ...) ------------------------------------------------------------------------
class _NormAct(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, gamma, beta, C, groups, act, slope, bn=None):
        x = x.contiguous()
        scale = shift = mean = rstd = None
        ctx.bn = None
        if bn is not None:
            scale, shift, mean, rstd, training, count = _bn_forward(x, C, bn)
            ctx.bn = (training, count)
        elif gamma is not None:
            scale, shift, mean, rstd = _gn_forward(x, C, groups, gamma.detach().float(), beta.detach().float())
        y = torch.empty_like(x)
        ops.norm_act_fwd(x, y, scale, shift, act, slope)
        ctx.save_for_backward(x, scale, shift, mean, rstd, None if gamma is None else gamma.detach().float())
        ctx.cfg = (C, groups, act, slope)
        return y

    @staticmethod
    def backward(ctx, g):
        x, scale, shift, mean, rstd, gamma = ctx.saved_tensors
        C, groups, act, slope = ctx.cfg
        g = g.contiguous()
        if act != ops.ACT_NONE:
            dz = torch.empty_like(g)
            ops.norm_act_bwd(g, x, dz, scale, shift, act, slope)
        else:
            dz = g
        if gamma is None:
            return dz, None, None, None, None, None, None, None
        if ctx.bn is not None:
            dx, dgamma, dbeta = _bn_backward(dz, x, C, gamma, mean, rstd, *ctx.bn)
        else:
            dx, dgamma, dbeta = _gn_backward(dz, x, C, groups, gamma, mean, rstd)
        return dx, dgamma, dbeta, None, None, None, None, None


def norm_act(a, gn=None, act=None, slope=None, bn=None):
    """act: None | 'r' | 'l' | 'e' (create_conv's letters); gn / bn: the normalisation in front of it (one fused pass)"""
    code, sl = (ops.ACT_NONE, 0.0) if act is None else ACT_CODES[act]
    if slope is not None:
        sl = slope
    gamma = beta = None
    groups = 1
    if gn is not None:
        gamma, beta, groups = gn.weight, gn.bias, gn.num_groups
        if gn.num_channels != a.C:
            raise MisError(f"GroupNorm over {gn.num_channels} channels applied to {a.C}")
    if bn is not None:
        gamma, beta = bn.weight, bn.bias
    if gn is None and bn is None and act is None:
        return a
    return CL(_NormAct.apply(a.t, gamma, beta, a.C, groups, code, sl, bn), a.C)


# ---- dropout ('d' = nn.Dropout, 'D' = nn.Dropout2d of create_conv, buildingblocks.py:105-109) -------------------------------------------------------------------
class _Dropout(torch.autograd.Function):
    """training-mode dropout: y = x * m / (1 - p).  The Bernoulli draws come from torch's generator on the device (torch.manual_seed governs them, as it governs the
    reference's; the reference's CPU stream itself is not reproduced: dropout parity is statistical), the arithmetic is mis_mask_scale / mis_norm_act_fwd.
    channelwise (Dropout2d on a 5-D input = torch's feature dropout): one draw per (sample, channel), applied as a per-(n, c) scale table."""

    @staticmethod
    def forward(ctx, x, C, p, channelwise):
        x = x.contiguous()
        N, Cp = x.shape[0], x.shape[-1]
        keep = 1.0 - p
        y = torch.empty_like(x)
        if channelwise:
            table = torch.zeros(N, Cp, device=x.device)
            table[:, :C] = torch.empty(N, C, device=x.device).bernoulli_(keep) / keep
            ops.norm_act_fwd(x, y, table, torch.zeros_like(table), ops.ACT_NONE)
            ctx.save_for_backward(table)
        else:
            mask = torch.empty(x.shape, dtype=x.dtype, device=x.device).bernoulli_(keep)
            ops.mask_scale(x, mask, y, 1.0 / keep)
            ctx.save_for_backward(mask)
        ctx.cfg = (channelwise, keep)
        return y

    @staticmethod
    def backward(ctx, g):
        (m,) = ctx.saved_tensors
        channelwise, keep = ctx.cfg
        g = g.contiguous()
        dx = torch.empty_like(g)
        if channelwise:
            ops.norm_act_fwd(g, dx, m, torch.zeros_like(m), ops.ACT_NONE)
        else:
            ops.mask_scale(g, m, dx, 1.0 / keep)
        return dx, None, None, None


def dropout(a, p, training, channelwise=False):
    if not training or p <= 0.0:
        return a
    if p >= 1.0:
        raise MisError("dropout with p = 1 zeroes everything: refuse")
    return CL(_Dropout.apply(a.t, a.C, float(p), channelwise), a.C)


# ---- pooling -------------------------------------------------------------------------------------------------------------------------------------
def _triple(k):
    return (k, k, k) if isinstance(k, int) else tuple(k)


class _Pool(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, k, avg):
        x = x.contiguous()
        N, D, H, W, Cp = x.shape
        kd, kh, kw = k
        if D < kd or H < kh or W < kw:
            raise MisError(f"pooling window {k} is larger than the {D}x{H}x{W} grid")
        y = torch.empty(N, D // kd, H // kh, W // kw, Cp, dtype=x.dtype, device=x.device)
        ops.pool3d_fwd(x, y, k, avg)
        ctx.save_for_backward(x)
        ctx.cfg = (k, avg)
        return y

    @staticmethod
    def backward(ctx, g):
        (x,) = ctx.saved_tensors
        k, avg = ctx.cfg
        dx = torch.empty_like(x)
        ops.pool3d_bwd(x, g.contiguous(), dx, k, avg)
        return dx, None, None


def pool(a, kernel_size=2, avg=False):
    return CL(_Pool.apply(a.t, _triple(kernel_size), avg), a.C)


# ---- nearest resize + concatenation (Decoder joining) -------------------------------------------------------------------------------------------------
class _ResizeCat(torch.autograd.Function):
    """out[..., :c0] = skip[..., :c0]; out[..., c0:c0+c1] = nearest-resize(low)[..., :c1] on skip's grid (skip may be None: plain resize to `size`)"""

    @staticmethod
    def forward(ctx, skip, low, c0, c1, size):
        low = low.contiguous()
        N = low.shape[0]
        src = tuple(low.shape[1:4])
        dev = low.device
        out = torch.zeros(N, *size, pad64(c0 + c1), dtype=low.dtype, device=dev)
        fwd, inv = ops.nearest_maps(src, size, dev)
        ops.gather3d_fwd(View(low, 0, c1), View(out, c0, c1), fwd)
        if skip is not None:
            skip = skip.contiguous()
            ident, _ = ops.nearest_maps(size, size, dev)
            ops.gather3d_fwd(View(skip, 0, c0), View(out, 0, c0), ident)
        ctx.cfg = (c0, c1, size, src, low.shape[-1], None if skip is None else skip.shape[-1])
        ctx.inv = inv
        return out

    @staticmethod
    def backward(ctx, g):
        c0, c1, size, src, lowp, skipp = ctx.cfg
        g = g.contiguous()
        N, dev = g.shape[0], g.device
        dlow = torch.zeros(N, *src, lowp, dtype=g.dtype, device=dev)
        ops.gather3d_bwd(View(g, c0, c1), View(dlow, 0, c1), ctx.inv)
        dskip = None
        if skipp is not None:
            dskip = torch.zeros(N, *size, skipp, dtype=g.dtype, device=dev)
            ident, _ = ops.nearest_maps(size, size, dev)
            ops.gather3d_fwd(View(g, 0, c0), View(dskip, 0, c0), ident)
        return dskip, dlow, None, None, None


def resize_nearest(a, size):
    size = tuple(int(s) for s in size)
    if a.grid[1:] == size:
        return a
    return CL(_ResizeCat.apply(None, a.t, 0, a.C, size), a.C)


def concat_resized(skip, low):
    """torch.cat((encoder_features, F.interpolate(x, size=encoder_features.size()[2:], mode='nearest')), dim=1)   (buildingblocks.py:548, :671-673)"""
    return CL(_ResizeCat.apply(skip.t, low.t, skip.C, low.C, skip.grid[1:]), skip.C + low.C)


# ---- sum joining / residual add ------------------------------------------------------------------------------------------------------------------------
class _AddAct(torch.autograd.Function):
    """y = act(a + b): `out += residual; out = non_linearity(out)` (buildingblocks.py:318-323) and the ResNet decoders' summation joining (:546)"""

    @staticmethod
    def forward(ctx, a, b, act, slope):
        a, b = a.contiguous(), b.contiguous()
        s = torch.empty_like(a)
        ops.add_act(a, b, s, relu=False)
        if act == ops.ACT_NONE:
            ctx.act = act
            return s
        y = torch.empty_like(a)
        ops.norm_act_fwd(s, y, None, None, act, slope)
        ctx.save_for_backward(s)
        ctx.act, ctx.slope = act, slope
        return y

    @staticmethod
    def backward(ctx, g):
        g = g.contiguous()
        if ctx.act == ops.ACT_NONE:
            return g, g, None, None
        (s,) = ctx.saved_tensors
        dz = torch.empty_like(g)
        ops.norm_act_bwd(g, s, dz, None, None, ctx.act, ctx.slope)
        return dz, dz, None, None


def add_act(a, b, act=None, slope=None):
    if a.C != b.C or a.grid != b.grid:
        raise MisError(f"sum joining of {a.grid + (a.C,)} and {b.grid + (b.C,)}")
    code, sl = (ops.ACT_NONE, 0.0) if act is None else ACT_CODES[act]
    if slope is not None:
        sl = slope
    return CL(_AddAct.apply(a.t, b.t, code, sl), a.C)


# ---- squeeze & excitation layers on their own (se.py:18-116) -----------------------------------------------------------------------------------------------
def _se_widths_ok(Cp, dtype):
    """csrc/se3d.hip: a power-of-two group of lanes (at most 64) shares a voxel, 16 bytes per lane and step, up to 4 steps"""
    nch = Cp // (8 if dtype == torch.bfloat16 else 4)
    G = min(nch, 64)
    return (G & (G - 1)) == 0 and nch % G == 0 and nch // G <= 4


class _SE(torch.autograd.Function):
    """y = max(x * cSE gate, x * sSE gate) (mode 0), x * cSE gate (1) or x * sSE gate (2).  fc weights come in their reference shapes ([C/r][C], [C][C/r]) and are
    zero-padded to the square [Cp][Cp] matrices of the kernels here; their gradients are cut back."""

    @staticmethod
    def forward(ctx, x, C, mode, fc1_w, fc1_b, fc2_w, fc2_b, conv_w, conv_b):
        x = x.contiguous()
        Cp = x.shape[-1]
        dev = x.device
        W1 = b1 = W2 = b2 = w = b0 = None
        if mode != ops.SE_SSE:
            Cr = fc1_w.shape[0]
            W1, W2 = torch.zeros(Cp, Cp, device=dev), torch.zeros(Cp, Cp, device=dev)
            b1, b2 = torch.zeros(Cp, device=dev), torch.zeros(Cp, device=dev)
            W1[:Cr, :C], W2[:C, :Cr], b1[:Cr], b2[:C] = fc1_w.detach().float(), fc2_w.detach().float(), fc1_b.detach().float(), fc2_b.detach().float()
            ctx.Cr = Cr
        if mode != ops.SE_CSE:
            w = torch.zeros(Cp, device=dev)
            w[:C] = conv_w.detach().float().reshape(-1)
            b0 = conv_b.detach().float().reshape(1).contiguous()
        y = torch.empty_like(x)
        state = ops.se_layer_fwd(x, y, mode, W1, b1, W2, b2, w, b0)
        ctx.save_for_backward(x, *[t for t in (W1, W2, w) if t is not None], *[t for t in state if t is not None])
        ctx.mode, ctx.C, ctx.wshape = mode, C, (None if conv_w is None else tuple(conv_w.shape))
        return y

    @staticmethod
    def backward(ctx, g):
        mode, C = ctx.mode, ctx.C
        sv = list(ctx.saved_tensors)
        x = sv.pop(0)
        W1 = W2 = w = mean = z1 = a = bgate = None
        if mode != ops.SE_SSE:
            W1, W2 = sv.pop(0), sv.pop(0)
        if mode != ops.SE_CSE:
            w = sv.pop(0)
        if mode != ops.SE_SSE:
            mean, z1, a = sv.pop(0), sv.pop(0), sv.pop(0)
        if mode != ops.SE_CSE:
            bgate = sv.pop(0)
        g = g.contiguous()
        dx = torch.empty_like(g)
        dW1, db1, dW2, db2, dw, db0 = ops.se_layer_bwd(g, x, dx, mode, (mean, z1, a, bgate), W1, W2, w)
        if mode != ops.SE_SSE:
            Cr = ctx.Cr
            dW1, db1, dW2, db2 = dW1[:Cr, :C].contiguous(), db1[:Cr].contiguous(), dW2[:C, :Cr].contiguous(), db2[:C].contiguous()
        if mode != ops.SE_CSE:
            dw, db0 = dw[:C].reshape(ctx.wshape), db0.reshape(1)
        return dx, None, None, dW1, db1, dW2, db2, dw, db0


def se(a, mode, cse=None, sse=None):
    """a: CL activation; cse: a ChannelSELayer3D (fc1, fc2), sse: a SpatialSELayer3D (conv) - whichever the mode uses"""
    if not _se_widths_ok(a.t.shape[-1], a.t.dtype):
        raise MisError(f"squeeze & excitation kernels: {a.C} channels (stored as {a.t.shape[-1]}) - the stored width must be 64 * 2^k or a multiple of "
                       f"{512 if a.t.dtype == torch.bfloat16 else 256} up to {2048 if a.t.dtype == torch.bfloat16 else 1024}")
    if mode != ops.SE_SSE and cse.fc1.weight.shape[1] != a.C or mode != ops.SE_CSE and sse.conv.weight.shape[1] != a.C:
        raise MisError(f"squeeze & excitation layer built for another channel count than the input's {a.C}")
    f1w, f1b, f2w, f2b = (cse.fc1.weight, cse.fc1.bias, cse.fc2.weight, cse.fc2.bias) if mode != ops.SE_SSE else (None,) * 4
    cw, cb = (sse.conv.weight, sse.conv.bias) if mode != ops.SE_CSE else (None, None)
    return CL(_SE.apply(a.t, a.C, mode, f1w, f1b, f2w, f2b, cw, cb), a.C)


# ---- ConvTranspose3d(k3, s2, p1, bias=False) + nearest resize to the encoder grid (TransposeConvUpsampling, buildingblocks.py:676-728) -------------------
class _ConvT2x(torch.autograd.Function):
    """low (N, d, h, w, Cinp) -> (N, 2d, 2h, 2w, Coutp): the (2d-1)^3 transposed-conv output resized (nearest) to exactly twice the input grid - the case
    the fused engine's column kernels (csrc/convt3d.hip) implement: one Cin -> 27*Cout GEMM, then a gather"""

    @staticmethod
    def forward(ctx, low, weight, cin, cout):
        low = low.contiguous()
        dev, dt = low.device, low.dtype
        N, d, h, w, cinp = low.shape
        coutp = pad64(cout)
        if tuple(weight.shape) != (cin, cout, 3, 3, 3):
            raise MisError(f"transposed conv: weight {tuple(weight.shape)} != ({cin}, {cout}, 3, 3, 3)")
        w2d = torch.zeros(27 * coutp, cinp, 1, device=dev)                       # row k*Coutp + co
        w2d.view(27, coutp, cinp)[:, :cout, :cin] = weight.detach().reshape(cin, cout, 27).permute(2, 1, 0)
        wf = torch.empty(1, 27 * coutp, cinp, dtype=dt, device=dev)
        wd = torch.empty(1, cinp, 27 * coutp, dtype=dt, device=dev)
        ops.pack_conv_weight(w2d, wf, wd)
        cols = torch.empty(N, d, h, w, 27 * coutp, dtype=dt, device=dev)
        ops.conv_igemm(low, wf, cols, ksize=1, Cin=cinp, Cout=27 * coutp)
        up = torch.empty(N, 2 * d, 2 * h, 2 * w, coutp, dtype=dt, device=dev)
        ops.convt3_col2im(cols, up)
        ctx.save_for_backward(low, wd)
        ctx.cfg = (cin, cout)
        return up

    @staticmethod
    def backward(ctx, g):
        low, wd = ctx.saved_tensors
        cin, cout = ctx.cfg
        g = g.contiguous()
        N, d, h, w, cinp = low.shape
        coutp = g.shape[-1]
        dev = low.device
        gcols = torch.empty(N, d, h, w, 27 * coutp, dtype=g.dtype, device=dev)
        ops.convt3_im2col(g, gcols)
        dw2d = torch.empty(27 * coutp, cinp, device=dev)
        ops.wgrad(low, gcols, dw2d, ksize=1, Cin=cinp, Cout=27 * coutp)
        dw = dw2d.view(27, coutp, cinp)[:, :cout, :cin].permute(2, 1, 0).reshape(cin, cout, 3, 3, 3).contiguous()
        dlow = None
        if ctx.needs_input_grad[0]:
            dlow = torch.empty_like(low)
            ops.conv_igemm(gcols, wd, dlow, ksize=1, Cin=27 * coutp, Cout=cinp)
        return dlow, dw, None, None


def conv_transpose_2x(a, weight, size):
    if weight.dim() == 4:
        # ConvTranspose2d(k3, s2, p1) of the is3d=False blocks on a depth-1 volume: the 3-D operator with the 2-D filter as its centre depth slice gives exactly that
        # at output depth 0 (z = 2*iz - 1 + kd admits only kd = 1), and the kernels' nearest resize (2d-1 -> 2d per axis) duplicates it into depth 1: keep depth 0
        if a.grid[1] != 1 or size[0] != 1:
            raise MisError(f"transposed conv: a ConvTranspose2d weight needs depth-1 activations, got {a.grid[1:]} -> {tuple(size)}")
        w3 = torch.nn.functional.pad(weight.unsqueeze(2), (0, 0, 0, 0, 1, 1))
        up = conv_transpose_2x(a, w3, (2,) + tuple(size[1:]))
        return CL(up.t[:, 0:1], up.C)
    cin, cout = weight.shape[:2]
    if cin != a.C:
        raise MisError(f"transposed conv: input has {a.C} channels, the weight expects {cin}")
    if tuple(size) != tuple(2 * s for s in a.grid[1:]):
        raise MisError(f"transposed-conv upsampling is built for an encoder grid of exactly twice the input grid: {a.grid[1:]} -> {tuple(size)}")
    return CL(_ConvT2x.apply(a.t, weight, cin, cout), cout)


# ---- create_conv order strings -------------------------------------------------------------------------------------------------------------------------
def run_single_conv(mod, a):
    """mod: a SingleConv container (children named as create_conv names them, `mod.order` = the order string).  Fusions: 'g' / 'b' directly in front of 'c'
    is folded into the convolution, 'r' directly behind 'c' is its epilogue, 'g' / 'b' + non-linearity behind the convolution is one pass."""
    order = mod.order
    i = 0
    while i < len(order):
        ch = order[i]
        nxt = order[i + 1] if i + 1 < len(order) else ""
        if ch in "gb" and nxt == "c":
            relu = order[i + 2:i + 3] == "r"
            kw = dict(gn=mod.groupnorm) if ch == "g" else dict(bn=mod.batchnorm)
            a = conv(a, mod.conv.weight, mod.conv.bias, relu=relu, **kw)
            i += 3 if relu else 2
        elif ch == "c":
            relu = nxt == "r"
            a = conv(a, mod.conv.weight, mod.conv.bias, relu=relu)
            i += 2 if relu else 1
        elif ch in "gb":
            act = nxt if nxt in ACT_CODES else None
            kw = dict(gn=mod.groupnorm) if ch == "g" else dict(bn=mod.batchnorm)
            a = norm_act(a, act=act, **kw)
            i += 2 if act else 1
        elif ch in ACT_CODES:
            a = norm_act(a, act=ch)
            i += 1
        elif ch in "dD":
            a = dropout(a, getattr(mod, "dropout_prob", 0.0), mod.training, channelwise=(ch == "D"))
            i += 1
        else:
            raise NotImplementedError(f"layer order character '{ch}' is not built on MI355X (built: g, b, c, r, l, e, d, D)")
    return a
