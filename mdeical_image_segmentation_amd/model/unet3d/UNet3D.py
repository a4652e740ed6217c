"""Mirror of model/unet3d/UNet3D.py:18-154 (HF wrapper). The reference file cannot even be imported as shipped
(`from model import UNet3D` fails); its semantics are restated from the source, including the quirk that the
ACTIVATED output is fed to the loss (sigmoid before BCEDiceLoss, which applies sigmoid again) - §8a-13q."""
from dataclasses import dataclass
from typing import Optional

import torch
from torch import nn
from transformers import PretrainedConfig, PreTrainedModel
from transformers.utils import ModelOutput

from ..._lib import MisError, check, load
from ...ops import stream_ptr
from .losses import get_loss_criterion
from .model import UNet3D


class _HipSigmoid(torch.autograd.Function):
    """the wrapper's `nn.Sigmoid()` on the logits (UNet3D.py:50, applied :140-141) as a HIP pass and its backward dL/dx = g * y * (1 - y) as another
    (mis_scale_sigmoid with a unit gate, csrc/pool_up.hip) - round 4 left this one elementwise op of call stack (D) to ATen"""

    @staticmethod
    def forward(ctx, x):
        if x.device.type != "cuda":
            raise MisError(f"UNet3DForMedicalSegmentation runs on MI355X only: logits on {x.device}")
        x = x.float().contiguous()
        one = torch.ones(1, device=x.device)
        y = torch.empty_like(x)
        check(load().mis_scale_sigmoid(x.data_ptr(), None, one.data_ptr(), 1, x.numel(), y.data_ptr(), stream_ptr()), "mis_scale_sigmoid")
        ctx.save_for_backward(x, one)
        return y

    @staticmethod
    def backward(ctx, g):
        x, one = ctx.saved_tensors
        g = g.float().contiguous()
        dx = torch.empty_like(x)
        check(load().mis_scale_sigmoid(x.data_ptr(), g.data_ptr(), one.data_ptr(), 1, x.numel(), dx.data_ptr(), stream_ptr()), "mis_scale_sigmoid")
        return dx


class Sigmoid(nn.Sigmoid):
    """nn.Sigmoid in the module tree (no parameters, same repr / state dict), HIP arithmetic"""

    def forward(self, x):
        return _HipSigmoid.apply(x)


class UNet3DForMedicalSegmentationConfig(PretrainedConfig):
    def __init__(self, unet_type="UNet3D", in_channels=1, out_channels=1, final_sigmoid=True, f_maps=64, layer_order="gcr", num_groups=8,
                 num_levels=4, is_segmentation=True, conv_padding=1, conv_upscale=2, upsample="default", dropout_prob=0.1,
                 loss_config=None, compute_dtype=None, **kwargs):
        super().__init__(**kwargs)
        self.unet_type, self.in_channels, self.out_channels, self.final_sigmoid = unet_type, in_channels, out_channels, final_sigmoid
        self.f_maps, self.layer_order, self.num_groups, self.num_levels = f_maps, layer_order, num_groups, num_levels
        self.is_segmentation, self.conv_padding, self.conv_upscale = is_segmentation, conv_padding, conv_upscale
        self.upsample, self.dropout_prob = upsample, dropout_prob
        self.loss_config = loss_config if loss_config is not None else {"loss": {"name": "BCEDiceLoss", "alpha": 1.0, "beta": 1.0}}
        self.compute_dtype = compute_dtype
        self.main_input_name = "volume"


@dataclass
class UNet3DForMedicalSegmentationOutput(ModelOutput):
    loss: Optional[torch.FloatTensor] = None
    logits: Optional[torch.FloatTensor] = None
    labels: Optional[torch.LongTensor] = None


class UNet3DForMedicalSegmentation(PreTrainedModel):
    config_class = UNet3DForMedicalSegmentationConfig
    main_input_name = "volume"

    def __init__(self, config):
        super().__init__(config)
        if config.unet_type != "UNet3D":
            raise NotImplementedError(f"unet_type={config.unet_type!r}: only UNet3D is built")
        self.model = UNet3D(in_channels=config.in_channels, out_channels=config.out_channels, final_sigmoid=config.final_sigmoid,
                            f_maps=config.f_maps, layer_order=config.layer_order, num_groups=config.num_groups, num_levels=config.num_levels,
                            is_segmentation=config.is_segmentation, conv_padding=config.conv_padding, conv_upscale=config.conv_upscale,
                            upsample=config.upsample, dropout_prob=config.dropout_prob, compute_dtype=config.compute_dtype)
        if config.is_segmentation and config.final_sigmoid:
            self.activation = Sigmoid()
        elif config.is_segmentation:
            self.activation = nn.Softmax(dim=1)
        else:
            self.activation = None
        self.loss_criterion = get_loss_criterion({"loss": dict(config.loss_config["loss"])})
        self.post_init()          # transformers >= 5: needed by from_pretrained; _init_weights is a no-op

    def _init_weights(self, module):
        return

    def forward(self, volume, target=None, weight=None):
        out = self.model(volume)
        activated = self.activation(out) if self.activation is not None else out
        loss = None
        if target is not None:
            loss = self.loss_criterion(activated, target) if weight is None else self.loss_criterion(activated, target, weight)
        return UNet3DForMedicalSegmentationOutput(loss=loss, logits=activated, labels=target)
