"""Mirror of the reference's model/unet2d/unet.py: `UNet` (:42-128) and the HuggingFace wrapper
`UNetConfig` / `UNetModel` / `UNetModelOutput` (:1156-1213), executed by the fused MI355X engine.

The nn.Module tree, parameter names and shapes are the reference's (stock nn.Conv2d / nn.ConvTranspose2d
containers: checkpoints round-trip with the reference), but `UNet.forward` does not call the sub-modules: all 46
parameters alias the engine's flat fp32 master buffer and one autograd.Function runs the whole network forward
(and, in backward, the whole backward) through libmisamd.
"""
import os
from dataclasses import dataclass

import torch
from torch import nn
from transformers import PretrainedConfig, PreTrainedModel
from transformers.utils import ModelOutput

from ... import ops
from ..._lib import MisError, check, load, stream_ptr
from ...engine2d import UNet2DEngine
from .init_weights import init_weights
from .layers import (CropAndConcat, DoubleConvolution, DownSample, UpSample, _BilinearUp, _Conv3x3Plain, _MaxPool2, _CatSlices, _MaxPoolCeil, conv_bn_relu, up_conv_bn_relu,
                     unetConv2)


def _dtype_from(name):
    name = (name or os.environ.get("MISAMD_DTYPE", "f32")).lower()
    if name in ("bf16", "bfloat16"):
        return torch.bfloat16
    if name in ("f32", "fp32", "float32"):
        return torch.float32
    raise MisError(f"compute dtype must be 'f32' or 'bf16', got {name!r}")


class _FusedUNet(torch.autograd.Function):
    """One autograd node for the whole network.  The engine keeps ONE set of activations, so a backward is only valid for the engine's
    LATEST forward: every forward stamps a generation counter and backward raises MisError on a mismatch (e.g. `l1 = m(a); l2 = m(b);
    (l1 + l2).backward()`, or an eval forward between a forward and its backward) instead of silently using the wrong activations.
    Gradients arriving through `logits` (auxiliary losses, custom compute_loss) are honoured: dL/dlogits of the fused criterion, scaled by
    g_loss, plus g_logits goes through the head kernel's external-gradient entry (`mis_head_loss`, LOSS_EXTERNAL)."""

    @staticmethod
    def forward(ctx, images, labels, owner, train, *params):
        eng = owner._engine_for(images)
        owner._sync_params_to_engine()
        loss, logits, _ = eng.forward(images.contiguous().float(), labels, train=train)
        eng.fwd_gen = getattr(eng, "fwd_gen", 0) + 1
        ctx.gen = eng.fwd_gen
        ctx.owner = owner
        ctx.train = train and labels is not None
        ctx.labels = labels
        ctx.set_materialize_grads(False)         # an unused output arrives as None, not as a tensor of zeros
        out_loss = loss.clone().reshape(()) if loss is not None else images.new_zeros(())
        out_logits = logits.clone()
        # (no copy of the logits is kept for the rare external-gradient path of backward: the generation check there guarantees that the engine's own logits buffer
        #  still holds THIS forward - ADVICE r2: a tensor stored on ctx as a plain attribute made a reference cycle through grad_fn)
        return out_loss, out_logits

    @staticmethod
    def backward(ctx, g_loss, g_logits):
        owner = ctx.owner
        eng = owner._engine
        if getattr(eng, "fwd_gen", 0) != ctx.gen:
            raise MisError("backward of a UNet forward that is no longer the engine's latest one: the fused engine keeps a single set of "
                           "activations - call backward() before running the model again (or use one model call per loss)")
        if g_loss is not None and not ctx.train:
            raise MisError("backward through the UNet loss needs labels and grad mode (the loss is fused into the head kernel)")
        if g_logits is None:
            if g_loss is None:
                return (None,) * (4 + len(list(owner.parameters())))
            scale = g_loss
            eng.backward()
        else:
            # rare path: something besides the fused criterion reads `logits`
            d = g_logits.to(torch.float32)
            if g_loss is not None:
                lg, lb = eng.logits, ctx.labels
                if eng.cout > 1:      # CrossEntropyLoss(mean): (softmax - onehot) / (N*H*W)   (reference unet.py:1184-1188, :1208)
                    dl = torch.softmax(lg, 1)
                    dl.scatter_add_(1, lb.unsqueeze(1), torch.full_like(dl[:, :1], -1.0))
                    dl /= lb.numel()
                else:                 # BCEWithLogitsLoss(mean): (sigmoid - t) / numel
                    dl = (torch.sigmoid(lg) - lb) / lb.numel()
                d = d + g_loss * dl
            eng.head_backward(d.contiguous())
            scale = None
            eng.backward()
        # the engine's gradients live in ONE flat fp32 buffer: the chain-rule factor (or the copy autograd needs, since the buffer is reused by the next backward) is one
        # kernel over it, and every parameter's gradient is a view of the result (round 4: this was 46 small multiplies, ~0.5 ms of launches per step)
        flat = eng.flat.g * scale if scale is not None else eng.flat.g.clone()
        grads = [eng.flat._view(flat, name) if p.requires_grad else None for name, p in owner.named_parameters()]
        return (None, None, None, None, *grads)


class UNet(nn.Module):
    """Classic 4-level U-Net 64-128-256-512-1024 with a 1x1 head (reference unet.py:42-128)."""

    def __init__(self, in_channels: int, out_channels: int, compute_dtype=None):
        super().__init__()
        self.down_conv = nn.ModuleList([DoubleConvolution(i, o) for i, o in [(in_channels, 64), (64, 128), (128, 256), (256, 512)]])
        self.down_sample = nn.ModuleList([DownSample() for _ in range(4)])
        self.middle_conv = DoubleConvolution(512, 1024)
        self.up_sample = nn.ModuleList([UpSample(i, o) for i, o in [(1024, 512), (512, 256), (256, 128), (128, 64)]])
        self.up_conv = nn.ModuleList([DoubleConvolution(i, o) for i, o in [(1024, 512), (512, 256), (256, 128), (128, 64)]])
        self.concat = nn.ModuleList([CropAndConcat() for _ in range(4)])
        self.final_conv = nn.Conv2d(64, out_channels, kernel_size=1)
        self.in_channels, self.out_channels = in_channels, out_channels
        self._compute_dtype = compute_dtype
        self._engine = None

    # ---- engine plumbing ----
    def _engine_for(self, images):
        if images.device.type != "cuda":
            raise MisError(f"UNet runs on MI355X only: got input on {images.device} (no CPU fallback)")
        if self._engine is None or self._engine.device != images.device:
            eng = UNet2DEngine(self.in_channels, self.out_channels, dtype=_dtype_from(self._compute_dtype), device=images.device)
            # alias every nn.Parameter onto the engine's flat fp32 master buffer
            for name, p in self.named_parameters():
                eng.P[name].copy_(p.detach().to(device=images.device, dtype=torch.float32))
                p.data = eng.P[name]
            self._engine = eng
        return self._engine

    def _sync_params_to_engine(self):
        eng = self._engine
        for name, p in self.named_parameters():
            if p.data_ptr() != eng.P[name].data_ptr():          # e.g. after load_state_dict(assign=True) / .to()
                eng.P[name].copy_(p.detach().to(torch.float32))
                p.data = eng.P[name]
        # Refresh the packed operands from the fp32 master on every call (0.3 ms): Tensor._version cannot be used to
        # detect an optimizer step, torch's fused AdamW (HF Trainer's default) updates parameters without bumping it.
        eng.repack()

    def forward(self, images: torch.Tensor, labels: torch.Tensor = None, _train: bool = None):
        train = self.training if _train is None else _train
        loss, logits = _FusedUNet.apply(images, labels, self, train and labels is not None, *self.parameters())
        if labels is None:
            return logits
        return logits, loss


@dataclass
class UNetModelOutput(ModelOutput):
    loss: torch.FloatTensor = None
    logits: torch.FloatTensor = None
    labels: torch.LongTensor = None


class UNetConfig(PretrainedConfig):
    def __init__(self, in_channels=1, out_channels=1, unet_type="UNet", compute_dtype=None, **kwargs):
        """unet_type: "UNet" (fused engine), "UNet_3Plus" / "UNet_3Plus_DeepSup" (per-layer HIP path + SegmentationLoss kernels); the reference's UNetModel has no branch for the CGM class either (it is used on its own).
        compute_dtype: "f32" (default, the parity mode) or "bf16"; env MISAMD_DTYPE overrides the default."""
        super().__init__(**kwargs)
        self.in_channels = in_channels
        self.out_channels = out_channels
        self.unet_type = unet_type
        self.compute_dtype = compute_dtype
        self.label_names = "labels"
        self.main_input_name = "images"
        self.keys_to_ignore_at_inference = ["labels"]


class UNetModel(PreTrainedModel):
    config_class = UNetConfig
    main_input_name = "images"
    _no_split_modules = []

    def __init__(self, config: UNetConfig):
        super().__init__(config)
        if config.unet_type == "UNet":
            self.unet = UNet(config.in_channels, config.out_channels, compute_dtype=getattr(config, "compute_dtype", None))
            # kept for interface parity (reference unet.py:1184-1188); the loss itself is computed in the head kernel
            self.criterion = nn.CrossEntropyLoss() if config.out_channels > 1 else nn.BCEWithLogitsLoss()
        elif config.unet_type in ("UNet_3Plus", "UNet_3Plus_DeepSup"):
            from .loss import SegmentationLoss
            cls = UNet_3Plus if config.unet_type == "UNet_3Plus" else UNet_3Plus_DeepSup
            self.unet = cls(config.in_channels, config.out_channels)
            self.criterion = SegmentationLoss()
        else:
            raise NotImplementedError(f"unet_type={config.unet_type!r}: 'UNet', 'UNet_3Plus' and 'UNet_3Plus_DeepSup' are built")
        # transformers >= 5 needs post_init() for from_pretrained / tied-weight bookkeeping (the 4.40-era reference does not call it);
        # _init_weights is a no-op, so the seeded PyTorch / kaiming initialisation above is left untouched
        self.post_init()

    def _init_weights(self, module):   # PyTorch default init already applied by the containers (as in the reference)
        return

    def forward(self, images: torch.Tensor, labels: torch.Tensor = None, **kwargs):
        if self.config.unet_type != "UNet":                # per-layer HIP path + the fused SegmentationLoss kernels (reference :1199-1213)
            out = self.unet(images)
            if isinstance(out, tuple):                        # deep supervision: the loss is summed over the five maps, logits = d1
                loss = sum(self.criterion(d, labels) for d in out) if labels is not None else None
                return UNetModelOutput(loss=loss, logits=out[0], labels=labels)
            loss = self.criterion(out, labels) if labels is not None else None
            return UNetModelOutput(loss=loss, logits=out, labels=labels)
        if labels is None:
            logits = self.unet(images, None)
            return UNetModelOutput(loss=None, logits=logits, labels=None)
        logits, loss = self.unet(images, labels, _train=torch.is_grad_enabled())
        return UNetModelOutput(loss=loss, logits=logits, labels=labels)


_FILTERS = [64, 128, 256, 512, 1024]


class _UNet3PlusBase(nn.Module):
    """UNet 3+ (reference model/unet2d/unet.py:136-446 and, with deep supervision, :454-787): 5 `unetConv2` encoder blocks; every decoder
    stage d = 4..1 fuses five 64-channel branches - encoder maps h_i (i < d) through MaxPool2d(2^(d-i), ceil_mode=True), h_d itself, and the
    deeper decoder maps hd_j (j > d) through bilinear up-sampling by 2^(j-d) - each through Conv3x3 + BatchNorm + ReLU, concatenated (320
    channels) and fused by another Conv3x3 + BN + ReLU; 3x3 convs produce the logits (one, or one per decoder stage up-sampled to full
    resolution).  Same module names, registration order (hence state-dict keys and seeded kaiming init) as the reference; every operator runs
    on the HIP kernels through the per-layer autograd functions of layers.py."""

    _deep_supervision = False
    _cgm = False

    def __init__(self, in_channels=3, n_classes=1, feature_scale=4, is_deconv=True, is_batchnorm=True):
        super().__init__()
        self.is_deconv, self.in_channels, self.is_batchnorm, self.feature_scale = is_deconv, in_channels, is_batchnorm, feature_scale
        f = _FILTERS
        for i in range(5):
            setattr(self, f"conv{i + 1}", unetConv2(in_channels if i == 0 else f[i - 1], f[i], is_batchnorm))
            if i < 4:
                setattr(self, f"maxpool{i + 1}", nn.MaxPool2d(kernel_size=2))
        self.CatChannels, self.CatBlocks = f[0], 5
        self.UpChannels = self.CatChannels * self.CatBlocks
        for d in (4, 3, 2, 1):
            for i in range(1, 6):
                name, cin = self._branch(d, i)
                if i < d:
                    setattr(self, name, nn.MaxPool2d(2 ** (d - i), 2 ** (d - i), ceil_mode=True))
                elif i > d:
                    setattr(self, name, nn.Upsample(scale_factor=2 ** (i - d), mode="bilinear"))
                setattr(self, name + "_conv", nn.Conv2d(cin, self.CatChannels, 3, padding=1))
                setattr(self, name + "_bn", nn.BatchNorm2d(self.CatChannels))
                setattr(self, name + "_relu", nn.ReLU(inplace=True))
            setattr(self, f"conv{d}d_1", nn.Conv2d(self.UpChannels, self.UpChannels, 3, padding=1))
            setattr(self, f"bn{d}d_1", nn.BatchNorm2d(self.UpChannels))
            setattr(self, f"relu{d}d_1", nn.ReLU(inplace=True))
        if self._deep_supervision:
            for k in (6, 5, 4, 3, 2):                                   # reference :643-648 (upscore6 is registered but never used)
                setattr(self, f"upscore{k}", nn.Upsample(scale_factor=2 ** (k - 1), mode="bilinear"))
            for k in range(1, 6):
                setattr(self, f"outconv{k}", nn.Conv2d(f[4] if k == 5 else self.UpChannels, n_classes, 3, padding=1))
        else:
            self.outconv1 = nn.Conv2d(self.UpChannels, n_classes, 3, padding=1)
        if self._cgm:                                                   # reference :998-1003
            self.cls = nn.Sequential(nn.Dropout(p=0.5), nn.Conv2d(f[4], 2, 1), nn.AdaptiveMaxPool2d(1), nn.Sigmoid())
        for m in self.modules():
            if isinstance(m, (nn.Conv2d, nn.BatchNorm2d)):
                init_weights(m, init_type="kaiming")

    def _branch(self, d, i):
        if i < d:
            return f"h{i}_PT_hd{d}", _FILTERS[i - 1]
        if i == d:
            return f"h{i}_Cat_hd{d}", _FILTERS[i - 1]
        return f"hd{i}_UT_hd{d}", (_FILTERS[4] if i == 5 else self.UpChannels)

    def _decode(self, inputs):
        if inputs.device.type != "cuda":
            raise MisError(f"{type(self).__name__} runs on MI355X only: got input on {inputs.device} (no CPU fallback)")
        h = {1: self.conv1(inputs)}
        for i in range(2, 6):
            h[i] = getattr(self, f"conv{i}")(_MaxPool2.apply(h[i - 1]))
        hd = {5: h[5]}
        for d in (4, 3, 2, 1):
            # the five 64-channel branches write straight into their slices of one NHWC buffer: the reference's torch.cat is free
            N, _, Hd, Wd = h[d].shape
            buf = torch.empty(N, Hd, Wd, self.UpChannels, dtype=h[d].dtype, device=inputs.device)
            parts = []
            for i in range(1, 6):
                name, _ = self._branch(d, i)
                out = ops.View(buf, self.CatChannels * (i - 1), self.CatChannels)
                if i > d:           # bilinear upsample -> conv -> BN -> ReLU, contracted at the low resolution (layers._UpConv3x3BNReLU)
                    parts.append(up_conv_bn_relu(hd[i], 2 ** (i - d), getattr(self, name + "_conv"), getattr(self, name + "_bn"), self.training, out))
                    continue
                src = _MaxPoolCeil.apply(h[i], 2 ** (d - i)) if i < d else h[i]
                parts.append(conv_bn_relu(src, getattr(self, name + "_conv"), getattr(self, name + "_bn"), self.training, out))
            cat = _CatSlices.apply(buf.permute(0, 3, 1, 2), *parts)
            hd[d] = conv_bn_relu(cat, getattr(self, f"conv{d}d_1"), getattr(self, f"bn{d}d_1"), self.training)
        return hd

    def _head(self, hd, k):
        conv = getattr(self, f"outconv{k}")
        return _Conv3x3Plain.apply(hd[k], conv.weight, conv.bias)


class UNet_3Plus(_UNet3PlusBase):
    def forward(self, inputs):
        return self._head(self._decode(inputs), 1)


class UNet_3Plus_DeepSup(_UNet3PlusBase):
    """five logit maps d1..d5 at full resolution (reference :768-787): outconv_k on decoder stage k, bilinear up-sampling by 2^(k-1)"""
    _deep_supervision = True

    def forward(self, inputs):
        hd = self._decode(inputs)
        outs = [self._head(hd, 1)]
        for k in range(2, 6):
            outs.append(_BilinearUp.apply(self._head(hd, k), 2 ** (k - 1)))
        return tuple(outs)


class _GateSigmoid(torch.autograd.Function):
    """sigmoid(d * gate[n]) for one deep-supervision map d (N, n_classes, H, W) - reference dotProduct + F.sigmoid (:1012-1017, 1147-1153); the gate is
    an arg-max (no gradient), so only d receives one"""

    @staticmethod
    def forward(ctx, d, gate):
        x = d.float().contiguous()
        out = torch.empty_like(x)
        check(load().mis_scale_sigmoid(x.data_ptr(), None, gate.data_ptr(), x.shape[0], x[0].numel(), out.data_ptr(), stream_ptr()), "mis_scale_sigmoid")
        ctx.save_for_backward(x, gate)
        return out

    @staticmethod
    def backward(ctx, gy):
        x, gate = ctx.saved_tensors
        g = gy.float().contiguous()
        gx = torch.empty_like(x)
        check(load().mis_scale_sigmoid(x.data_ptr(), g.data_ptr(), gate.data_ptr(), x.shape[0], x[0].numel(), gx.data_ptr(), stream_ptr()), "mis_scale_sigmoid")
        return gx, None


class UNet_3Plus_DeepSup_CGM(_UNet3PlusBase):
    """UNet 3+ with deep supervision and the classification-guided module (reference :795-1153): a 2-way classifier on the deepest feature map
    (Dropout -> Conv1x1 -> global max -> sigmoid -> arg-max) gates all five maps, which are returned as PROBABILITIES sigmoid(d_k * gate).  The
    classifier is evaluated by `mis_cgm_gate`; its arg-max carries no gradient (the reference trains it with no loss either).  n_classes must be 1
    (the reference's einsum 'ijk,ij->ijk' needs it)."""
    _deep_supervision = True
    _cgm = True

    def __init__(self, in_channels=3, n_classes=1, feature_scale=4, is_deconv=True, is_batchnorm=True):
        if n_classes != 1:
            raise MisError("UNet_3Plus_DeepSup_CGM: n_classes must be 1 (the class gate is broadcast over a single segmentation channel)")
        super().__init__(in_channels, n_classes, feature_scale, is_deconv, is_batchnorm)
        self.last_cls = None                            # (N, 2) classifier scores of the last forward

    def forward(self, inputs):
        hd = self._decode(inputs)
        x = hd[5]
        if self.training:                               # nn.Dropout(p=0.5) of the classifier branch (torch's generator, like the reference)
            x = torch.nn.functional.dropout(x, 0.5, True)
        conv = self.cls[1]
        N, C, h, w = x.shape
        xv = x.permute(0, 2, 3, 1)
        if not xv.is_contiguous():
            xv = xv.contiguous()
        cls = torch.empty(N, 2, dtype=torch.float32, device=x.device)
        gate = torch.empty(N, dtype=torch.float32, device=x.device)
        check(load().mis_cgm_gate(ops.dtype_code(xv.dtype), xv.data_ptr(), C, N, h * w, C, conv.weight.detach().float().contiguous().data_ptr(),
                                  conv.bias.detach().float().contiguous().data_ptr(), cls.data_ptr(), gate.data_ptr(), stream_ptr()), "mis_cgm_gate")
        self.last_cls = cls
        outs = [self._head(hd, 1)]
        for k in range(2, 6):
            outs.append(_BilinearUp.apply(self._head(hd, k), 2 ** (k - 1)))
        return tuple(_GateSigmoid.apply(d, gate) for d in outs)
