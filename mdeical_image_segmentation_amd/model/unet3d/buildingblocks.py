"""Mirror of model/unet3d/buildingblocks.py: `create_conv` order strings, SingleConv, DoubleConv, ResNetBlock(SE), Encoder, Decoder and the upsampling
modules, with the reference's module tree (state-dict keys are identical:
`encoders.{i}.basic_module.SingleConv{j}.{groupnorm.weight,groupnorm.bias,conv.weight}`).

Two execution routes share these parameter containers:
  * the whole-network fused engines (engine3d*.py, selected by model.py for the configurations they cover) read the parameters directly;
  * every block is ALSO callable on its own, `block(x)` with x = (N, C, D, H, W) on the GPU, exactly like the reference's modules: the arithmetic is the chain of
    HIP-kernel autograd functions of `blocks3d.py` (MFMA implicit-GEMM convolution with the GroupNorm folded in, generic pooling / nearest-resize / activation
    passes).  model.py composes the same `_cl` methods for configurations the fused engines refuse (general channel counts, grids, pooling, layer orders).
There is no CPU path: a tensor that is not on the GPU raises."""
import torch
from torch import nn

from ... import blocks3d as B
from .se import ChannelSELayer3D, ChannelSpatialSELayer3D, SpatialSELayer3D


def create_conv(in_channels, out_channels, kernel_size, order, num_groups, padding, dropout_prob, is3d):
    """buildingblocks.py:14-113: the (name, module) list of one conv layer for an order string over 'c' conv, 'g' groupnorm, 'r' ReLU, 'l' LeakyReLU, 'e' ELU,
    'b' batchnorm, 'd' / 'D' dropout.  The conv has a bias only without a norm in the order (:62)."""
    assert "c" in order, "Conv layer MUST be present"
    assert order[0] not in "rle", "Non-linearity cannot be the first operation in the layer"
    if not ((kernel_size == 3 and padding == 1) or (kernel_size == 1 and padding == 0)):
        raise NotImplementedError("the MI355X convolution kernels are built for kernel 3 / padding 1 and kernel 1 / padding 0")
    mods = []
    for i, char in enumerate(order):
        if char == "r":
            mods.append(("ReLU", nn.ReLU(inplace=True)))
        elif char == "l":
            mods.append(("LeakyReLU", nn.LeakyReLU(inplace=True)))
        elif char == "e":
            mods.append(("ELU", nn.ELU(inplace=True)))
        elif char == "c":
            bias = not ("g" in order or "b" in order)
            mods.append(("conv", (nn.Conv3d if is3d else nn.Conv2d)(in_channels, out_channels, kernel_size, padding=padding, bias=bias)))     # (:65-68)
        elif char == "g":
            num_channels = in_channels if i < order.index("c") else out_channels
            groups = 1 if num_channels < num_groups else num_groups
            assert num_channels % groups == 0, \
                f"Expected number of channels in input to be divisible by num_groups. num_channels={num_channels}, num_groups={groups}"
            mods.append(("groupnorm", nn.GroupNorm(num_groups=groups, num_channels=num_channels)))
        elif char == "d":
            mods.append(("dropout", nn.Dropout(p=dropout_prob)))
        elif char == "D":
            mods.append(("dropout2d", nn.Dropout2d(p=dropout_prob)))
        elif char == "b":
            mods.append(("batchnorm", (nn.BatchNorm3d if is3d else nn.BatchNorm2d)(in_channels if i < order.index("c") else out_channels)))      # (:97-104)
        else:
            raise ValueError(f"Unsupported layer type '{char}'. MUST be one of ['b', 'g', 'r', 'l', 'e', 'c', 'd', 'D']")
    return mods


class _Block:
    """forward(x): (N, C, D, H, W) fp32 on the GPU (is3d=False blocks: (N, C, H, W), carried as a depth-1 volume) -> the same layout, through the channels-last HIP
    route (`_cl`)"""

    def forward(self, x):
        return B.from_cl(self._cl(B.to_cl(x)), two_d=x.dim() == 4)


class SingleConv(_Block, nn.Sequential):
    """buildingblocks.py:116-159"""

    def __init__(self, in_channels, out_channels, kernel_size=3, order="gcr", num_groups=8, padding=1, dropout_prob=0.1, is3d=True):
        super().__init__()
        self.order, self.dropout_prob = order, dropout_prob
        for name, module in create_conv(in_channels, out_channels, kernel_size, order, num_groups, padding, dropout_prob, is3d):
            self.add_module(name, module)

    def _cl(self, a):
        return B.run_single_conv(self, a)


class DoubleConv(_Block, nn.Sequential):
    """buildingblocks.py:162-252: encoder in -> max(in, out//2) -> out ; decoder in -> out -> out."""

    def __init__(self, in_channels, out_channels, encoder, kernel_size=3, order="gcr", num_groups=8, padding=1, upscale=2,
                 dropout_prob=0.1, is3d=True):
        super().__init__()
        if encoder:
            c1_in = in_channels
            c1_out = out_channels if upscale == 1 else out_channels // 2
            if c1_out < in_channels:
                c1_out = in_channels
            c2_in, c2_out = c1_out, out_channels
        else:
            c1_in, c1_out = in_channels, out_channels
            c2_in, c2_out = out_channels, out_channels
        # the reference accepts a pair of dropout probabilities, one per conv (:228-232)
        p1, p2 = dropout_prob if isinstance(dropout_prob, (list, tuple)) else (dropout_prob, dropout_prob)
        self.add_module("SingleConv1", SingleConv(c1_in, c1_out, kernel_size, order, num_groups, padding, p1, is3d))
        self.add_module("SingleConv2", SingleConv(c2_in, c2_out, kernel_size, order, num_groups, padding, p2, is3d))

    def _cl(self, a):
        return self.SingleConv2._cl(self.SingleConv1._cl(a))


class ResNetBlock(_Block, nn.Module):
    """buildingblocks.py:255-325: conv1 = 1x1x1 conv (with bias) when the channel count changes, conv2 = SingleConv(order), conv3 = SingleConv without the
    non-linearity, which follows the residual add (LeakyReLU(0.1) for an 'l' order, ELU for 'e', else ReLU).  The reference's default order is 'cge'."""

    def __init__(self, in_channels, out_channels, kernel_size=3, order="cge", num_groups=8, is3d=True, **kwargs):
        super().__init__()
        self.conv1 = (nn.Conv3d if is3d else nn.Conv2d)(in_channels, out_channels, 1) if in_channels != out_channels else nn.Identity()     # (:262-268)
        self.conv2 = SingleConv(out_channels, out_channels, kernel_size=kernel_size, order=order, num_groups=num_groups, is3d=is3d)
        n_order = order
        for c in "rel":
            n_order = n_order.replace(c, "")
        self.conv3 = SingleConv(out_channels, out_channels, kernel_size=kernel_size, order=n_order, num_groups=num_groups, is3d=is3d)
        if "l" in order:
            self.non_linearity = nn.LeakyReLU(negative_slope=0.1, inplace=True)
            self._act = ("l", 0.1)
        elif "e" in order:
            self.non_linearity = nn.ELU(inplace=True)
            self._act = ("e", 1.0)
        else:
            self.non_linearity = nn.ReLU(inplace=True)
            self._act = ("r", 0.0)

    def _cl(self, a):
        residual = a if isinstance(self.conv1, nn.Identity) else B.conv(a, self.conv1.weight, self.conv1.bias)
        out = self.conv3._cl(self.conv2._cl(residual))
        return B.add_act(out, residual, act=self._act[0], slope=self._act[1])


class ResNetBlockSE(ResNetBlock):
    """buildingblocks.py:326-362: ResNetBlock followed by a squeeze-and-excitation module with reduction_ratio 1: 'scse' ChannelSpatialSELayer3D, 'cse'
    ChannelSELayer3D, 'sse' SpatialSELayer3D (csrc/se3d.hip).  Inside ResidualUNetSE3D ('scse', the only one Encoder / Decoder pass) the fused residual engine
    runs the block; called on its own it is the stand-alone chain of `_cl` functions."""

    def __init__(self, in_channels, out_channels, kernel_size=3, order="cge", num_groups=8, se_module="scse", **kwargs):
        super().__init__(in_channels, out_channels, kernel_size=kernel_size, order=order, num_groups=num_groups, **kwargs)
        assert se_module in ["scse", "cse", "sse"]
        if se_module == "scse":
            self.se_module = ChannelSpatialSELayer3D(num_channels=out_channels, reduction_ratio=1)
        elif se_module == "cse":
            self.se_module = ChannelSELayer3D(num_channels=out_channels, reduction_ratio=1)
        else:
            self.se_module = SpatialSELayer3D(num_channels=out_channels)

    def _cl(self, a):
        return self.se_module._cl(super()._cl(a))


class Encoder(_Block, nn.Module):
    """buildingblocks.py:365-439: optional MaxPool3d / AvgPool3d(pool_kernel_size) then the basic module"""

    def __init__(self, in_channels, out_channels, conv_kernel_size=3, apply_pooling=True, pool_kernel_size=2, pool_type="max",
                 basic_module=DoubleConv, conv_layer_order="gcr", num_groups=8, padding=1, upscale=2, dropout_prob=0.1, is3d=True):
        super().__init__()
        assert pool_type in ["max", "avg"]
        self._is3d = is3d
        if apply_pooling:                                    # (:409-418: MaxPool3d / AvgPool3d, or their 2-D twins)
            cls = (nn.MaxPool3d if pool_type == "max" else nn.AvgPool3d) if is3d else (nn.MaxPool2d if pool_type == "max" else nn.AvgPool2d)
            self.pooling = cls(kernel_size=pool_kernel_size)
        else:
            self.pooling = None
        self.basic_module = basic_module(in_channels, out_channels, encoder=True, kernel_size=conv_kernel_size, order=conv_layer_order,
                                         num_groups=num_groups, padding=padding, upscale=upscale, dropout_prob=dropout_prob, is3d=is3d)

    def _cl(self, a):
        if self.pooling is not None:
            k = self.pooling.kernel_size
            if not self._is3d:                               # a 2-D window on the depth-1 volume
                k = (1,) + ((k, k) if isinstance(k, int) else tuple(k))
            a = B.pool(a, k, avg=isinstance(self.pooling, (nn.AvgPool3d, nn.AvgPool2d)))
        return self.basic_module._cl(a)


class AbstractUpsampling(nn.Module):
    """buildingblocks.py:626-644: forward(encoder_features, x) upsamples x to the encoder features' spatial size"""

    def forward(self, encoder_features, x):
        size = tuple(encoder_features.shape[2:])
        two_d = x.dim() == 4
        return B.from_cl(self._cl(B.to_cl(x), (1,) + size if two_d else size), two_d=two_d)


class InterpolateUpsampling(AbstractUpsampling):
    def __init__(self, mode="nearest"):
        super().__init__()
        if mode != "nearest":
            raise NotImplementedError("only nearest upsampling is built")

    def _cl(self, a, size):
        return B.resize_nearest(a, size)


class TransposeConvUpsampling(AbstractUpsampling):
    """buildingblocks.py:676-728: ConvTranspose3d(k, stride=scale, padding=1, bias=False) + F.interpolate(size) (`upsample.conv_transposed.weight`)"""

    class Upsample(nn.Module):
        def __init__(self, conv_transposed, is3d):
            super().__init__()
            self.conv_transposed = conv_transposed
            self.is3d = is3d

        def forward(self, x, size):
            two_d = x.dim() == 4
            size = (1,) + tuple(size) if two_d else tuple(size)
            return B.from_cl(B.conv_transpose_2x(B.to_cl(x), self.conv_transposed.weight, size), two_d=two_d)

    def __init__(self, in_channels, out_channels, kernel_size=3, scale_factor=2, is3d=True):
        super().__init__()
        if kernel_size != 3 or scale_factor != 2:
            raise NotImplementedError("only ConvTranspose3d / ConvTranspose2d(k3, s2, p1) upsampling is built")
        ct = nn.ConvTranspose3d if is3d else nn.ConvTranspose2d          # (:700-718)
        self.upsample = self.Upsample(ct(in_channels, out_channels, kernel_size=3, stride=2, padding=1, bias=False), is3d)

    def _cl(self, a, size):
        return B.conv_transpose_2x(a, self.upsample.conv_transposed.weight, size)


class NoUpsampling(AbstractUpsampling):
    def _cl(self, a, size):
        return a


class Decoder(nn.Module):
    """buildingblocks.py:442-550: upsampling to the encoder features' grid, joining (concat for DoubleConv, sum for the ResNet blocks), basic module"""

    def __init__(self, in_channels, out_channels, conv_kernel_size=3, scale_factor=2, basic_module=DoubleConv, conv_layer_order="gcr",
                 num_groups=8, padding=1, upsample="default", dropout_prob=0.1, is3d=True):
        super().__init__()
        concat, adapt_channels = True, False
        if upsample is not None and upsample != "none":
            if upsample == "default":
                if basic_module is DoubleConv:
                    upsample, concat, adapt_channels = "nearest", True, False
                elif basic_module in (ResNetBlock, ResNetBlockSE):
                    upsample, concat, adapt_channels = "deconv", False, True
            if upsample == "deconv":
                self.upsampling = TransposeConvUpsampling(in_channels, out_channels, kernel_size=conv_kernel_size, scale_factor=scale_factor, is3d=is3d)
            else:
                self.upsampling = InterpolateUpsampling(mode=upsample)
        else:
            self.upsampling = NoUpsampling()
        self._concat = concat
        if adapt_channels:
            in_channels = out_channels
        self.basic_module = basic_module(in_channels, out_channels, encoder=False, kernel_size=conv_kernel_size, order=conv_layer_order,
                                         num_groups=num_groups, padding=padding, dropout_prob=dropout_prob, is3d=is3d)

    def _cl(self, skip, a):
        size = skip.grid[1:]
        if self._concat and isinstance(self.upsampling, InterpolateUpsampling):
            joined = B.concat_resized(skip, a)                          # resize + concat in one pass
        else:
            up = self.upsampling._cl(a, size)
            joined = B.concat_resized(skip, up) if self._concat else B.add_act(skip, up)
        return self.basic_module._cl(joined)

    def forward(self, encoder_features, x):
        return B.from_cl(self._cl(B.to_cl(encoder_features), B.to_cl(x)), two_d=x.dim() == 4)


def create_encoders(in_channels, f_maps, basic_module, conv_kernel_size, conv_padding, conv_upscale, dropout_prob, layer_order, num_groups,
                    pool_kernel_size, is3d):
    encoders = []
    for i, out in enumerate(f_maps):
        encoders.append(Encoder(in_channels if i == 0 else f_maps[i - 1], out, apply_pooling=(i != 0), basic_module=basic_module,
                                conv_layer_order=layer_order, conv_kernel_size=conv_kernel_size, num_groups=num_groups,
                                pool_kernel_size=pool_kernel_size, padding=conv_padding, upscale=conv_upscale, dropout_prob=dropout_prob,
                                is3d=is3d))
    return nn.ModuleList(encoders)


def create_decoders(f_maps, basic_module, conv_kernel_size, conv_padding, layer_order, num_groups, upsample, dropout_prob, is3d):
    decoders = []
    rf = list(reversed(f_maps))
    for i in range(len(rf) - 1):
        in_feature_num = rf[i] + rf[i + 1] if (basic_module is DoubleConv and upsample != "deconv") else rf[i]
        decoders.append(Decoder(in_feature_num, rf[i + 1], basic_module=basic_module, conv_layer_order=layer_order,
                                conv_kernel_size=conv_kernel_size, num_groups=num_groups, padding=conv_padding, upsample=upsample,
                                dropout_prob=dropout_prob, is3d=is3d))
    return nn.ModuleList(decoders)
