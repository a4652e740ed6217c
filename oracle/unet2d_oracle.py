"""ORACLE (test infrastructure, not product code) - 2-D U-Net train step on the CPU.

A plain functional restatement of the reference's hot path in stock PyTorch CPU
ops (the reference's own arithmetic is ATen: every op below is the op the
reference's nn.Module calls).  Only tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg may import this file.  It is pinned against golden
vectors generated from the real reference import (tests/golden/make_golden.py,
tests/test_oracle_vs_golden.py).

Reference lines restated:
  DoubleConvolution  model/unet2d/layers.py:103-133  (conv3x3 p1 bias -> ReLU, twice; no norm)
  DownSample         model/unet2d/layers.py:136-150  (MaxPool2d(2))
  UpSample           model/unet2d/layers.py:153-168  (ConvTranspose2d k2 s2, weight [Cin,Cout,2,2])
  CropAndConcat      model/unet2d/layers.py:171-192  (center_crop(skip) ; cat([x, skip], 1))
  UNet               model/unet2d/unet.py:42-128
  UNetModel.forward  model/unet2d/unet.py:1199-1213  (CE if out_channels>1 else BCEWithLogits)
  CustomTrainer      trainer/MYtrainer.py:6-11       (loss = model(**inputs)["loss"])
  HF Trainer step    clip_grad_norm_(1.0) -> AdamW(b1 .9, b2 .999, eps 1e-8, wd on non-bias params)
"""
import math

import torch
import torch.nn.functional as F

FEATS = [64, 128, 256, 512]


def param_specs(in_channels, out_channels):
    """(name, shape) in the reference's registration order (model/unet2d/unet.py:47-89)."""
    specs = []

    def dc(prefix, ci, co):
        specs.append((f"{prefix}.first.weight", (co, ci, 3, 3)))
        specs.append((f"{prefix}.first.bias", (co,)))
        specs.append((f"{prefix}.second.weight", (co, co, 3, 3)))
        specs.append((f"{prefix}.second.bias", (co,)))

    for i, (ci, co) in enumerate([(in_channels, 64), (64, 128), (128, 256), (256, 512)]):
        dc(f"down_conv.{i}", ci, co)
    dc("middle_conv", 512, 1024)
    for i, (ci, co) in enumerate([(1024, 512), (512, 256), (256, 128), (128, 64)]):
        specs.append((f"up_sample.{i}.up.weight", (ci, co, 2, 2)))
        specs.append((f"up_sample.{i}.up.bias", (co,)))
    for i, (ci, co) in enumerate([(1024, 512), (512, 256), (256, 128), (128, 64)]):
        dc(f"up_conv.{i}", ci, co)
    specs.append(("final_conv.weight", (out_channels, 64, 1, 1)))
    specs.append(("final_conv.bias", (out_channels,)))
    return specs


def _default_conv_init_(w, b):
    """torch.nn.modules.conv._ConvNd.reset_parameters: kaiming_uniform_(a=sqrt(5)) then
    bias ~ U(-1/sqrt(fan_in), 1/sqrt(fan_in)); fan_in = weight.size(1) * receptive field."""
    torch.nn.init.kaiming_uniform_(w, a=math.sqrt(5))
    if b is not None:
        fan_in = w.size(1) * (w[0][0].numel() if w.dim() > 2 else 1)
        bound = 1.0 / math.sqrt(fan_in) if fan_in > 0 else 0
        torch.nn.init.uniform_(b, -bound, bound)


def init_params(in_channels, out_channels, seed=0, dtype=torch.float32):
    """Same RNG consumption order as `torch.manual_seed(seed); UNet(in,out)` in the reference."""
    torch.manual_seed(seed)
    specs = param_specs(in_channels, out_channels)
    params = {}
    i = 0
    while i < len(specs):
        (wn, ws), (bn, bs) = specs[i], specs[i + 1]
        w = torch.empty(ws, dtype=dtype)
        b = torch.empty(bs, dtype=dtype)
        _default_conv_init_(w, b)
        params[wn], params[bn] = w, b
        i += 2
    return params


def double_conv(x, p, prefix):
    x = F.relu(F.conv2d(x, p[f"{prefix}.first.weight"], p[f"{prefix}.first.bias"], padding=1))
    x = F.relu(F.conv2d(x, p[f"{prefix}.second.weight"], p[f"{prefix}.second.bias"], padding=1))
    return x


def center_crop(t, h, w):
    H, W = t.shape[-2], t.shape[-1]
    top = int(round((H - h) / 2.0))
    left = int(round((W - w) / 2.0))
    return t[..., top:top + h, left:left + w]


def unet_forward(p, images):
    """model/unet2d/unet.py:95-128."""
    x = images
    skips = []
    for i in range(4):
        x = double_conv(x, p, f"down_conv.{i}")
        skips.append(x)
        x = F.max_pool2d(x, 2)
    x = double_conv(x, p, "middle_conv")
    for i in range(4):
        x = F.conv_transpose2d(x, p[f"up_sample.{i}.up.weight"], p[f"up_sample.{i}.up.bias"], stride=2)
        s = center_crop(skips.pop(), x.shape[2], x.shape[3])
        x = torch.cat([x, s], dim=1)
        x = double_conv(x, p, f"up_conv.{i}")
    return F.conv2d(x, p["final_conv.weight"], p["final_conv.bias"])


def criterion(logits, labels):
    """model/unet2d/unet.py:1184-1188: CrossEntropyLoss() if C>1 else BCEWithLogitsLoss()."""
    if logits.shape[1] > 1:
        return F.cross_entropy(logits, labels)
    return F.binary_cross_entropy_with_logits(logits, labels)


def model_forward(p, images, labels):
    logits = unet_forward(p, images)
    loss = criterion(logits, labels) if labels is not None else None
    return loss, logits


def loss_and_grads(p, images, labels):
    ps = {k: v.detach().clone().requires_grad_(True) for k, v in p.items()}
    loss, logits = model_forward(ps, images, labels)
    loss.backward()
    grads = {k: v.grad.detach() for k, v in ps.items()}
    return loss.detach(), logits.detach(), grads


# ---- bf16-STORAGE emulation (test infrastructure for the engine's bf16 mode) -----------------------------------------------
# The engine's bf16 mode stores activations, activation gradients and the packed conv weights in bf16 and accumulates in fp32; fp32 stay the
# image, the first layer's and the head's weights, every weight gradient and the master weights.  The functions below restate exactly that on top
# of the fp32 oracle (round-to-nearest-even at the same tensor boundaries), so the bf16 engine can be held to a tight bar: on this net the
# bf16-storage gradients differ from the fp32 reference by 3-12 % (relative L2) by construction - measured with this emulation, no device
# code involved - which a comparison against the fp32 oracle alone cannot tell from a wrong kernel.
class _RoundAct(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        return x.bfloat16().float()

    @staticmethod
    def backward(ctx, g):
        return g.bfloat16().float()


class _RoundWeight(torch.autograd.Function):
    @staticmethod
    def forward(ctx, w):
        return w.bfloat16().float()

    @staticmethod
    def backward(ctx, g):
        return g


def unet_forward_bf16_storage(p, images):
    ra, rw = _RoundAct.apply, _RoundWeight.apply

    def dc(x, prefix, first_fp32=False):
        w1 = p[f"{prefix}.first.weight"] if first_fp32 else rw(p[f"{prefix}.first.weight"])
        x = ra(F.relu(F.conv2d(x, w1, p[f"{prefix}.first.bias"], padding=1)))
        return ra(F.relu(F.conv2d(x, rw(p[f"{prefix}.second.weight"]), p[f"{prefix}.second.bias"], padding=1)))

    x = images
    skips = []
    for i in range(4):
        x = dc(x, f"down_conv.{i}", first_fp32=(i == 0))
        skips.append(x)
        x = F.max_pool2d(x, 2)
    x = dc(x, "middle_conv")
    for i in range(4):
        x = ra(F.conv_transpose2d(x, rw(p[f"up_sample.{i}.up.weight"]), p[f"up_sample.{i}.up.bias"], stride=2))
        s = center_crop(skips.pop(), x.shape[2], x.shape[3])
        x = dc(torch.cat([x, s], dim=1), f"up_conv.{i}")
    return F.conv2d(x, p["final_conv.weight"], p["final_conv.bias"])


def loss_and_grads_bf16_storage(p, images, labels):
    ps = {k: v.detach().clone().requires_grad_(True) for k, v in p.items()}
    logits = unet_forward_bf16_storage(ps, images)
    loss = criterion(logits, labels)
    loss.backward()
    return loss.detach(), logits.detach(), {k: v.grad.detach() for k, v in ps.items()}


def clip_grad_norm(grads, max_norm=1.0):
    """torch.nn.utils.clip_grad_norm_: total = ||(||g_i||_2)_i||_2 ; coef = clamp(max/(total+1e-6), max=1)."""
    norms = torch.stack([g.norm(2) for g in grads.values()])
    total = norms.norm(2)
    coef = torch.clamp(max_norm / (total + 1e-6), max=1.0)
    return total, {k: g * coef for k, g in grads.items()}


def is_decayed(name):
    """HF Trainer.get_decay_parameter_names: everything except biases (no LayerNorm in this net)."""
    return not name.endswith("bias")


class AdamW:
    """torch.optim.AdamW single-tensor semantics (decoupled weight decay, bias correction)."""

    def __init__(self, params, lr=5e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-3):
        self.lr, self.b1, self.b2, self.eps, self.wd = lr, betas[0], betas[1], eps, weight_decay
        self.m = {k: torch.zeros_like(v) for k, v in params.items()}
        self.v = {k: torch.zeros_like(v) for k, v in params.items()}
        self.t = 0

    def step(self, params, grads, lr=None):
        lr = self.lr if lr is None else lr
        self.t += 1
        bc1 = 1 - self.b1 ** self.t
        bc2 = 1 - self.b2 ** self.t
        for k, p in params.items():
            g = grads[k]
            if is_decayed(k):
                p.mul_(1 - lr * self.wd)
            self.m[k].lerp_(g, 1 - self.b1)
            self.v[k].mul_(self.b2).addcmul_(g, g, value=1 - self.b2)
            denom = (self.v[k].sqrt() / math.sqrt(bc2)).add_(self.eps)
            p.addcdiv_(self.m[k], denom, value=-lr / bc1)


def train_step(p, opt, images, labels, max_grad_norm=1.0, lr=None):
    """One HF-Trainer inner step: fwd, bwd, clip, AdamW (in place on p). Returns loss, grad-norm, logits."""
    loss, logits, grads = loss_and_grads(p, images, labels)
    total, grads = clip_grad_norm(grads, max_grad_norm)
    opt.step(p, grads, lr)
    return loss, total, logits


def argmax_mask(logits):
    """trainer/metrcis.py:159 (commented) / model/unet3d/predictor.py:167: channel argmax, lowest index on ties."""
    return logits.argmax(dim=1)


def unet_conv2(x, p, n=2, is_batchnorm=True, training=True, eps=1e-5, momentum=0.1):
    """`unetConv2` (reference model/unet2d/layers.py:8-46): n x [Conv2d(3, s1, p1, bias) -> BatchNorm2d -> ReLU].
    p: state-dict style {"conv1.0.weight", "conv1.0.bias", "conv1.1.weight", "conv1.1.bias", "conv1.1.running_mean", ...};
    in training mode the running statistics in p are REPLACED by their updated values (momentum 0.1, unbiased variance)."""
    for i in range(1, n + 1):
        x = F.conv2d(x, p[f"conv{i}.0.weight"], p[f"conv{i}.0.bias"], padding=1)
        if is_batchnorm:
            rm, rv = p[f"conv{i}.1.running_mean"].clone(), p[f"conv{i}.1.running_var"].clone()
            x = F.batch_norm(x, rm, rv, p[f"conv{i}.1.weight"], p[f"conv{i}.1.bias"], training, momentum, eps)
            if training:
                p[f"conv{i}.1.running_mean"], p[f"conv{i}.1.running_var"] = rm, rv
        x = F.relu(x)
    return x
