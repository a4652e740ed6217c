"""Mirror of model/unet3d/model.py: `AbstractUNet` / `UNet3D` / `ResidualUNet3D` / `ResidualUNetSE3D` (:13-280).

`forward(x)` returns LOGITS (the reference disabled the final activation, model.py:145-149) and is differentiable.  Two routes:
  * FUSED (engine3d*.py) for the configurations the whole-network engines cover - 1 input channel, f_maps = multiples of 64 starting at 64, layer order 'gcr',
    MaxPool3d(2), grids divisible by 2^(levels-1): one autograd.Function runs the fused forward; its backward feeds the external dL/dlogits to the head
    kernel and then the fused backward.  Parameters are stock containers aliased onto the engine's flat fp32 master buffer.
  * BLOCKS (blocks3d.py through buildingblocks.*._cl) for everything else the reference's constructor accepts: any channel counts, in_channels > 1, other
    pooling windows, odd grids (`F.interpolate(size=...)`), 'cge' / 'cl' / 'crg' ... layer orders; the reference's forward loop (model.py:121-150) over
    per-layer HIP autograd functions, channels-last from the first encoder to the final 1x1x1 convolution.
`MISAMD_3D_ROUTE=blocks` forces the second route (tests compare the two on the same weights); `=fused` makes an uncovered configuration an error."""
import os

import torch
from torch import nn

from ... import blocks3d as B
from ..._lib import MisError
from ...engine3d import UNet3DEngine
from ...engine3d_res import ResidualUNet3DEngine, ResidualUNetSE3DEngine
from .buildingblocks import DoubleConv, ResNetBlock, ResNetBlockSE, create_decoders, create_encoders
from .utils import number_of_features_per_level


def _dtype_from(name):
    name = (name or os.environ.get("MISAMD_DTYPE", "f32")).lower()
    if name in ("bf16", "bfloat16"):
        return torch.bfloat16
    if name in ("f32", "fp32", "float32"):
        return torch.float32
    raise MisError(f"compute dtype must be 'f32' or 'bf16', got {name!r}")


class _FusedUNet3D(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, owner, *params):
        eng = owner._engine_for(x)
        owner._sync_params_to_engine()
        _, logits, _ = eng.forward(x.contiguous().float(), None)
        eng.fwd_gen = getattr(eng, "fwd_gen", 0) + 1      # the engine keeps ONE set of activations: backward is valid for its latest forward only
        ctx.gen = eng.fwd_gen
        ctx.owner = owner
        return logits.clone()

    @staticmethod
    def backward(ctx, g):
        owner = ctx.owner
        eng = owner._engine
        if getattr(eng, "fwd_gen", 0) != ctx.gen:
            raise MisError("backward of a UNet3D forward that is no longer the engine's latest one: the fused engine keeps a single set of "
                           "activations - call backward() before running the model again")
        eng.head_backward(g.contiguous().float())
        eng.backward()
        grads = [eng.Gr[name].clone() if p.requires_grad else None for name, p in owner.named_parameters()]
        return (None, None, *grads)


class AbstractUNet(nn.Module):
    def __init__(self, in_channels, out_channels, final_sigmoid, basic_module, f_maps=64, layer_order="gcr", num_groups=8, num_levels=4,
                 is_segmentation=True, conv_kernel_size=3, pool_kernel_size=2, conv_padding=1, conv_upscale=2, upsample="default",
                 dropout_prob=0.1, is3d=True, compute_dtype=None):
        super().__init__()
        if isinstance(f_maps, int):
            f_maps = number_of_features_per_level(f_maps, num_levels=num_levels)
        assert isinstance(f_maps, (list, tuple)) and len(f_maps) > 1, "Required at least 2 levels in the U-Net"
        if basic_module not in (DoubleConv, ResNetBlock, ResNetBlockSE) or (not is3d and basic_module is ResNetBlockSE):
            raise NotImplementedError("U-Nets with DoubleConv (UNet3D / UNet2D), ResNetBlock (ResidualUNet3D / ResidualUNet2D) or ResNetBlockSE (3-D) basic modules are built")
        self._is3d = is3d
        self._residual = basic_module in (ResNetBlock, ResNetBlockSE)
        self._se = basic_module is ResNetBlockSE
        self.encoders = create_encoders(in_channels, f_maps, basic_module, conv_kernel_size, conv_padding, conv_upscale, dropout_prob,
                                        layer_order, num_groups, pool_kernel_size, is3d)
        self.decoders = create_decoders(f_maps, basic_module, conv_kernel_size, conv_padding, layer_order, num_groups, upsample,
                                        dropout_prob, is3d)
        self.final_conv = (nn.Conv3d if is3d else nn.Conv2d)(f_maps[0], out_channels, 1)      # (model.py:89-92)
        if is_segmentation:
            self.final_activation = nn.Sigmoid() if final_sigmoid else nn.Softmax(dim=1)
        else:
            self.final_activation = None
        self._cfg = (in_channels, out_channels, list(f_maps), num_groups, "deconv" if upsample == "deconv" else "default")
        ups_ok = upsample in ("default", "deconv") if self._residual else upsample in ("default", "nearest", "deconv")
        # what the whole-network engines cover (engine3d.UNet3DEngine.__init__ / engine3d_res): everything else takes the per-block route
        self._fused_cfg_ok = (is3d and in_channels == 1 and 1 <= out_channels <= 4 and f_maps[0] == 64 and all(f % 64 == 0 for f in f_maps) and
                              layer_order == "gcr" and conv_kernel_size == 3 and conv_padding == 1 and conv_upscale == 2 and
                              pool_kernel_size in (2, (2, 2, 2)) and ups_ok)
        self._compute_dtype = compute_dtype
        self._engine = None

    def _route(self, x):
        want = os.environ.get("MISAMD_3D_ROUTE", "auto").lower()
        if want not in ("auto", "fused", "blocks"):
            raise MisError(f"MISAMD_3D_ROUTE must be auto, fused or blocks, got {want!r}")
        div = 1 << (len(self.encoders) - 1)
        fused_ok = self._fused_cfg_ok and x.dim() == 5 and all(int(d) % div == 0 for d in x.shape[2:])
        if want == "fused" and not fused_ok:
            raise MisError("MISAMD_3D_ROUTE=fused, but this configuration / input grid is outside the fused 3-D engines (see model/unet3d/model.py)")
        if want == "blocks" or not fused_ok:
            if self._se:
                raise NotImplementedError("ResidualUNetSE3D runs on the fused engine only: 1 input channel, f_maps multiples of 64 from 64, order 'gcr', "
                                          f"grid divisible by {div}")
            return "blocks"
        return "fused"

    def _forward_blocks(self, x):
        """the reference's forward loop (model.py:121-150) over the channels-last HIP blocks; logits out"""
        if x.device.type != "cuda":
            raise MisError(f"UNet3D runs on MI355X only: got input on {x.device} (no CPU fallback)")
        if any(p.device != x.device for p in self.parameters()):
            self.to(x.device)
        if x.dim() != (5 if self._is3d else 4):
            raise MisError(f"{type(self).__name__} expects a {'(N, C, D, H, W)' if self._is3d else '(N, C, H, W)'} input, got {tuple(x.shape)}")
        a = B.to_cl(x, _dtype_from(self._compute_dtype))          # (a 2-D input travels as a depth-1 volume: blocks3d.to_cl)
        feats = []
        for enc in self.encoders:
            a = enc._cl(a)
            feats.insert(0, a)
        for dec, skip in zip(self.decoders, feats[1:]):
            a = dec._cl(skip, a)
        return B.from_cl(B.conv(a, self.final_conv.weight, self.final_conv.bias), two_d=not self._is3d)

    def _engine_for(self, x):
        if x.device.type != "cuda":
            raise MisError(f"UNet3D runs on MI355X only: got input on {x.device} (no CPU fallback)")
        if self._engine is None or self._engine.device != x.device:
            cin, cout, f_maps, groups, upsample = self._cfg
            if self._residual:
                eng = (ResidualUNetSE3DEngine if self._se else ResidualUNet3DEngine)(cin, cout, f_maps=f_maps, num_groups=groups, dtype=_dtype_from(self._compute_dtype), device=x.device)
            else:
                eng = UNet3DEngine(cin, cout, f_maps=f_maps, num_groups=groups, dtype=_dtype_from(self._compute_dtype), device=x.device,
                                   upsample=upsample)
            for name, p in self.named_parameters():
                eng.P[name].copy_(p.detach().to(device=x.device, dtype=torch.float32))
                p.data = eng.P[name]
            # the engine computed its GroupNorm conditioning flags on its OWN random init: redo them on the module's parameters, and wait (a checkpoint with
            # |gamma| < |beta| / 16 would otherwise take the statistics-from-dW route on its first steps)
            eng.repack()
            eng.refresh_gn_flags(sync=True)
            self._engine = eng
        return self._engine

    def _sync_params_to_engine(self):
        eng = self._engine
        for name, p in self.named_parameters():
            if p.data_ptr() != eng.P[name].data_ptr():
                eng.P[name].copy_(p.detach().to(torch.float32))
                p.data = eng.P[name]
        eng.repack()                                        # every call: see model/unet2d/unet.py

    def forward(self, x):
        if self._route(x) == "blocks":
            return self._forward_blocks(x)
        return _FusedUNet3D.apply(x, self, *self.parameters())


class UNet3D(AbstractUNet):
    def __init__(self, in_channels, out_channels, final_sigmoid=True, f_maps=64, layer_order="gcr", num_groups=8, num_levels=4,
                 is_segmentation=True, conv_padding=1, conv_upscale=2, upsample="default", dropout_prob=0.1, **kwargs):
        super().__init__(in_channels=in_channels, out_channels=out_channels, final_sigmoid=final_sigmoid, basic_module=DoubleConv,
                         f_maps=f_maps, layer_order=layer_order, num_groups=num_groups, num_levels=num_levels,
                         is_segmentation=is_segmentation, conv_padding=conv_padding, conv_upscale=conv_upscale, upsample=upsample,
                         dropout_prob=dropout_prob, is3d=True, compute_dtype=kwargs.get("compute_dtype"))


class ResidualUNet3D(AbstractUNet):
    """model.py:197-232: ResNetBlock basic module, summation joining, transposed-conv upsampling; num_levels defaults to 5"""

    def __init__(self, in_channels, out_channels, final_sigmoid=True, f_maps=64, layer_order="gcr", num_groups=8, num_levels=5,
                 is_segmentation=True, conv_padding=1, conv_upscale=2, upsample="default", dropout_prob=0.1, **kwargs):
        super().__init__(in_channels=in_channels, out_channels=out_channels, final_sigmoid=final_sigmoid, basic_module=ResNetBlock,
                         f_maps=f_maps, layer_order=layer_order, num_groups=num_groups, num_levels=num_levels,
                         is_segmentation=is_segmentation, conv_padding=conv_padding, conv_upscale=conv_upscale, upsample=upsample,
                         dropout_prob=dropout_prob, is3d=True, compute_dtype=kwargs.get("compute_dtype"))


class ResidualUNetSE3D(AbstractUNet):
    """model.py:235-280: ResidualUNet3D with `ResNetBlockSE` blocks (squeeze & excitation 'scse' after every residual block)"""

    def __init__(self, in_channels, out_channels, final_sigmoid=True, f_maps=64, layer_order="gcr", num_groups=8, num_levels=5,
                 is_segmentation=True, conv_padding=1, conv_upscale=2, upsample="default", dropout_prob=0.1, **kwargs):
        super().__init__(in_channels=in_channels, out_channels=out_channels, final_sigmoid=final_sigmoid, basic_module=ResNetBlockSE,
                         f_maps=f_maps, layer_order=layer_order, num_groups=num_groups, num_levels=num_levels,
                         is_segmentation=is_segmentation, conv_padding=conv_padding, conv_upscale=conv_upscale, upsample=upsample,
                         dropout_prob=dropout_prob, is3d=True, compute_dtype=kwargs.get("compute_dtype"))


class UNet2D(AbstractUNet):
    """model.py:283-320: the same encoder / decoder tree over Conv2d / MaxPool2d (is3d=False).  Runs on the per-block HIP route: images travel as depth-1 volumes,
    the 3x3 convolutions are the 2-D implicit-GEMM kernels on (N, H, W, C) views of them"""

    def __init__(self, in_channels, out_channels, final_sigmoid=True, f_maps=64, layer_order="gcr", num_groups=8, num_levels=4,
                 is_segmentation=True, conv_padding=1, conv_upscale=2, upsample="default", dropout_prob=0.1, **kwargs):
        super().__init__(in_channels=in_channels, out_channels=out_channels, final_sigmoid=final_sigmoid, basic_module=DoubleConv,
                         f_maps=f_maps, layer_order=layer_order, num_groups=num_groups, num_levels=num_levels,
                         is_segmentation=is_segmentation, conv_padding=conv_padding, conv_upscale=conv_upscale, upsample=upsample,
                         dropout_prob=dropout_prob, is3d=False, compute_dtype=kwargs.get("compute_dtype"))


class ResidualUNet2D(AbstractUNet):
    """model.py:323-359: ResNetBlock basic modules, sum joining, ConvTranspose2d(k3, s2, p1) upsampling, is3d=False"""

    def __init__(self, in_channels, out_channels, final_sigmoid=True, f_maps=64, layer_order="gcr", num_groups=8, num_levels=5,
                 is_segmentation=True, conv_padding=1, conv_upscale=2, upsample="default", dropout_prob=0.1, **kwargs):
        super().__init__(in_channels=in_channels, out_channels=out_channels, final_sigmoid=final_sigmoid, basic_module=ResNetBlock,
                         f_maps=f_maps, layer_order=layer_order, num_groups=num_groups, num_levels=num_levels,
                         is_segmentation=is_segmentation, conv_padding=conv_padding, conv_upscale=conv_upscale, upsample=upsample,
                         dropout_prob=dropout_prob, is3d=False, compute_dtype=kwargs.get("compute_dtype"))


def get_model(model_config):
    cfg = dict(model_config)
    name = cfg.pop("name")
    if name == "UNet3D":
        return UNet3D(**cfg)
    if name == "ResidualUNet3D":
        return ResidualUNet3D(**cfg)
    if name == "ResidualUNetSE3D":
        return ResidualUNetSE3D(**cfg)
    if name == "UNet2D":
        return UNet2D(**cfg)
    if name == "ResidualUNet2D":
        return ResidualUNet2D(**cfg)
    raise NotImplementedError(f"get_model: UNet3D, ResidualUNet3D, ResidualUNetSE3D, UNet2D and ResidualUNet2D are built, got {name}")
