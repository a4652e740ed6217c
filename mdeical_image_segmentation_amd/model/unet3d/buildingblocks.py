"""Mirror of model/unet3d/buildingblocks.py (the 'gcr' / DoubleConv / Encoder / Decoder subset of the hot path).

The classes are PARAMETER CONTAINERS with the reference's module tree (so state-dict keys are identical:
`encoders.{i}.basic_module.SingleConv{j}.{groupnorm.weight,groupnorm.bias,conv.weight}`); the arithmetic of a whole
network runs in engine3d.UNet3DEngine through one autograd.Function (see model.py). Calling a block on its own raises:
stand-alone blocks are not part of the accelerated path."""
import torch
from torch import nn

from .se import ChannelSpatialSELayer3D


def create_conv(in_channels, out_channels, kernel_size, order, num_groups, padding, dropout_prob, is3d):
    """buildingblocks.py:14-113 for order 'gcr' (GroupNorm -> Conv3d(no bias) -> ReLU) and 'gc' (the last conv of a ResNetBlock)."""
    if order not in ("gcr", "gc") or not is3d or kernel_size != 3 or padding != 1:
        raise NotImplementedError("only layer_order='gcr' (and its 'gc' tail), 3-D, kernel 3, padding 1 is built (SURVEY.md §8a-8)")
    if in_channels < num_groups:
        num_groups = 1
    assert in_channels % num_groups == 0
    mods = [("groupnorm", nn.GroupNorm(num_groups=num_groups, num_channels=in_channels)),
            ("conv", nn.Conv3d(in_channels, out_channels, kernel_size, padding=padding, bias=False))]
    if "r" in order:
        mods.append(("ReLU", nn.ReLU(inplace=True)))
    return mods


class _ContainerOnly:
    def forward(self, *a, **k):
        raise NotImplementedError(f"{type(self).__name__} is a parameter container here: run the whole UNet3D (fused MI355X engine)")


class SingleConv(_ContainerOnly, nn.Sequential):
    def __init__(self, in_channels, out_channels, kernel_size=3, order="gcr", num_groups=8, padding=1, dropout_prob=0.1, is3d=True):
        super().__init__()
        for name, module in create_conv(in_channels, out_channels, kernel_size, order, num_groups, padding, dropout_prob, is3d):
            self.add_module(name, module)


class DoubleConv(_ContainerOnly, nn.Sequential):
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
        self.add_module("SingleConv1", SingleConv(c1_in, c1_out, kernel_size, order, num_groups, padding, dropout_prob, is3d))
        self.add_module("SingleConv2", SingleConv(c2_in, c2_out, kernel_size, order, num_groups, padding, dropout_prob, is3d))


class ResNetBlock(_ContainerOnly, nn.Module):
    """buildingblocks.py:255-325 (order 'gcr'): conv1 = 1x1x1 conv when the channel count changes, conv2 = SingleConv(order), conv3 = SingleConv
    without the non-linearity, which follows the residual add.  Parameter container; computed by engine3d_res.ResidualUNet3DEngine."""

    def __init__(self, in_channels, out_channels, kernel_size=3, order="gcr", num_groups=8, is3d=True, **kwargs):
        super().__init__()
        if order != "gcr" or not is3d:
            raise NotImplementedError("ResNetBlock on MI355X: layer_order='gcr' (ResidualUNet3D's default), 3-D")
        self.conv1 = nn.Conv3d(in_channels, out_channels, 1) if in_channels != out_channels else nn.Identity()
        self.conv2 = SingleConv(out_channels, out_channels, kernel_size=kernel_size, order=order, num_groups=num_groups, is3d=is3d)
        n_order = order
        for c in "rel":
            n_order = n_order.replace(c, "")
        self.conv3 = SingleConv(out_channels, out_channels, kernel_size=kernel_size, order=n_order, num_groups=num_groups, is3d=is3d)
        self.non_linearity = nn.ReLU(inplace=True)


class ResNetBlockSE(ResNetBlock):
    """buildingblocks.py:326-362: ResNetBlock followed by a squeeze-and-excitation module.  Only se_module='scse' (the one ResidualUNetSE3D can reach:
    Encoder / Decoder never pass another) is built: ChannelSpatialSELayer3D with reduction_ratio 1 (csrc/se3d.hip)."""

    def __init__(self, in_channels, out_channels, kernel_size=3, order="cge", num_groups=8, se_module="scse", **kwargs):
        super().__init__(in_channels, out_channels, kernel_size=kernel_size, order=order, num_groups=num_groups, **kwargs)
        assert se_module in ["scse", "cse", "sse"]
        if se_module != "scse":
            raise NotImplementedError("ResNetBlockSE on MI355X: se_module='scse'")
        self.se_module = ChannelSpatialSELayer3D(num_channels=out_channels, reduction_ratio=1)


class Encoder(_ContainerOnly, nn.Module):
    def __init__(self, in_channels, out_channels, conv_kernel_size=3, apply_pooling=True, pool_kernel_size=2, pool_type="max",
                 basic_module=DoubleConv, conv_layer_order="gcr", num_groups=8, padding=1, upscale=2, dropout_prob=0.1, is3d=True):
        super().__init__()
        if pool_type != "max" or pool_kernel_size != 2:
            raise NotImplementedError("only MaxPool3d(2) is built")
        self.pooling = nn.MaxPool3d(kernel_size=2) if apply_pooling else None
        self.basic_module = basic_module(in_channels, out_channels, encoder=True, kernel_size=conv_kernel_size, order=conv_layer_order,
                                         num_groups=num_groups, padding=padding, upscale=upscale, dropout_prob=dropout_prob, is3d=is3d)


class InterpolateUpsampling(_ContainerOnly, nn.Module):
    def __init__(self, mode="nearest"):
        super().__init__()
        if mode != "nearest":
            raise NotImplementedError("only nearest upsampling is built")


class TransposeConvUpsampling(_ContainerOnly, nn.Module):
    """buildingblocks.py:676-728: ConvTranspose3d(k, stride=scale, padding=1, bias=False) + F.interpolate(size) - parameter container
    (`upsample.conv_transposed.weight`); computed by engine3d._ct_fwd / _ct_bwd."""

    class Upsample(_ContainerOnly, nn.Module):
        def __init__(self, conv_transposed, is3d):
            super().__init__()
            self.conv_transposed = conv_transposed
            self.is3d = is3d

    def __init__(self, in_channels, out_channels, kernel_size=3, scale_factor=2, is3d=True):
        super().__init__()
        if not is3d or kernel_size != 3 or scale_factor != 2:
            raise NotImplementedError("only ConvTranspose3d(k3, s2, p1) upsampling is built")
        self.upsample = self.Upsample(nn.ConvTranspose3d(in_channels, out_channels, kernel_size=3, stride=2, padding=1, bias=False), is3d)


class Decoder(_ContainerOnly, nn.Module):
    def __init__(self, in_channels, out_channels, conv_kernel_size=3, scale_factor=2, basic_module=DoubleConv, conv_layer_order="gcr",
                 num_groups=8, padding=1, upsample="default", dropout_prob=0.1, is3d=True):
        super().__init__()
        if basic_module in (ResNetBlock, ResNetBlockSE):
            # buildingblocks.py:486-534: 'default' -> transposed-conv upsampling, SUM joining, the block sees out_channels
            if upsample not in ("default", "deconv"):
                raise NotImplementedError("ResNetBlock decoders: transposed-conv upsampling + sum joining (the reference's default) is built")
            self.upsampling = TransposeConvUpsampling(in_channels, out_channels, kernel_size=conv_kernel_size, scale_factor=scale_factor, is3d=is3d)
            self.basic_module = basic_module(out_channels, out_channels, encoder=False, kernel_size=conv_kernel_size, order=conv_layer_order,
                                             num_groups=num_groups, padding=padding, dropout_prob=dropout_prob, is3d=is3d)
            return
        if upsample not in ("default", "nearest", "deconv") or basic_module is not DoubleConv:
            raise NotImplementedError("only DoubleConv decoders with nearest or 'deconv' upsampling + concat are built (SURVEY.md §8a-11)")
        if upsample == "deconv":
            self.upsampling = TransposeConvUpsampling(in_channels, out_channels, kernel_size=conv_kernel_size, scale_factor=scale_factor, is3d=is3d)
        else:
            self.upsampling = InterpolateUpsampling("nearest")
        self.basic_module = basic_module(in_channels, out_channels, encoder=False, kernel_size=conv_kernel_size, order=conv_layer_order,
                                         num_groups=num_groups, padding=padding, dropout_prob=dropout_prob, is3d=is3d)


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
