"""ORACLE (test infrastructure, not product code) - 3-D U-Net (pytorch-3dunet flavour) on the CPU.

Functional restatement in stock PyTorch CPU ops of:
  create_conv / SingleConv 'gcr'  model/unet3d/buildingblocks.py:14-159  (GroupNorm -> Conv3d(no bias) -> ReLU)
  DoubleConv                      model/unet3d/buildingblocks.py:162-252 (enc: in->max(in,out//2)->out; dec: in->out->out)
  Encoder                         model/unet3d/buildingblocks.py:365-439 (MaxPool3d(2) except level 0)
  Decoder / InterpolateUpsampling model/unet3d/buildingblocks.py:442-550,642-673 (nearest to encoder size; cat((enc, x),1))
  AbstractUNet.forward            model/unet3d/model.py:125-151 (returns logits)
  number_of_features_per_level    model/unet3d/utils.py:109-110
  BCEDiceLoss / DiceLoss / compute_per_channel_dice / flatten  model/unet3d/losses.py:7-33,83-129,167-178,258-270
  UNet3DForMedicalSegmentation.forward quirk (sigmoid before the loss)  model/unet3d/UNet3D.py:134-154
Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this file.
"""
import math

import torch
import torch.nn.functional as F


def number_of_features_per_level(init_channel_number, num_levels):
    return [init_channel_number * 2 ** k for k in range(num_levels)]


def _groups(num_channels, num_groups):
    return 1 if num_channels < num_groups else num_groups


def layer_plan(in_channels, f_maps=64, num_levels=4, upsample="default"):
    """Returns (encoders, decoders): lists of [(cin, cout), (cin, cout)] per DoubleConv.
    upsample='deconv' (buildingblocks.py:604-626): the decoder's DoubleConv sees f[i] channels (encoder f[i+1] + transposed-conv f[i+1])."""
    if isinstance(f_maps, int):
        f_maps = number_of_features_per_level(f_maps, num_levels)
    enc = []
    for i, out in enumerate(f_maps):
        cin = in_channels if i == 0 else f_maps[i - 1]
        c1 = out // 2
        if c1 < cin:
            c1 = cin
        enc.append([(cin, c1), (c1, out)])
    dec = []
    rf = list(reversed(f_maps))
    for i in range(len(rf) - 1):
        cin = rf[i] + rf[i + 1] if upsample != "deconv" else rf[i]
        out = rf[i + 1]
        dec.append([(cin, out), (out, out)])
    return enc, dec, f_maps


def param_specs(in_channels, out_channels, f_maps=64, num_levels=4, upsample="default"):
    enc, dec, f_maps = layer_plan(in_channels, f_maps, num_levels, upsample)
    rf = list(reversed(f_maps))
    specs = []
    for grp, plan in (("encoders", enc), ("decoders", dec)):
        for i, convs in enumerate(plan):
            if grp == "decoders" and upsample == "deconv":   # registered before basic_module (buildingblocks.py:504-534)
                specs.append((f"decoders.{i}.upsampling.upsample.conv_transposed.weight", (rf[i], rf[i + 1], 3, 3, 3)))
            for j, (ci, co) in enumerate(convs):
                pre = f"{grp}.{i}.basic_module.SingleConv{j + 1}"
                specs.append((f"{pre}.groupnorm.weight", (ci,)))
                specs.append((f"{pre}.groupnorm.bias", (ci,)))
                specs.append((f"{pre}.conv.weight", (co, ci, 3, 3, 3)))
    specs.append(("final_conv.weight", (out_channels, f_maps[0], 1, 1, 1)))
    specs.append(("final_conv.bias", (out_channels,)))
    return specs


def init_params(in_channels, out_channels, f_maps=64, num_levels=4, seed=0, upsample="default"):
    """Same RNG order as torch.manual_seed(seed); UNet3D(in, out, f_maps=...) (GroupNorm init draws nothing)."""
    torch.manual_seed(seed)
    p = {}
    for name, shape in param_specs(in_channels, out_channels, f_maps, num_levels, upsample):
        if name.endswith("groupnorm.weight"):
            p[name] = torch.ones(shape)
        elif name.endswith("groupnorm.bias"):
            p[name] = torch.zeros(shape)
        elif (name.endswith("conv.weight") or name.endswith("conv_transposed.weight")) and not name.startswith("final"):
            w = torch.empty(shape)
            torch.nn.init.kaiming_uniform_(w, a=math.sqrt(5))
            p[name] = w
        elif name == "final_conv.weight":
            w = torch.empty(shape)
            torch.nn.init.kaiming_uniform_(w, a=math.sqrt(5))
            p[name] = w
        elif name == "final_conv.bias":
            fan_in = shape and p["final_conv.weight"].size(1)
            b = torch.empty(shape)
            bound = 1.0 / math.sqrt(fan_in)
            torch.nn.init.uniform_(b, -bound, bound)
            p[name] = b
    return p


def single_conv(x, p, pre, num_groups=8):
    g = _groups(x.shape[1], num_groups)
    x = F.group_norm(x, g, p[f"{pre}.groupnorm.weight"], p[f"{pre}.groupnorm.bias"], eps=1e-5)
    x = F.conv3d(x, p[f"{pre}.conv.weight"], None, padding=1)
    return F.relu(x)


def double_conv(x, p, pre, num_groups=8):
    x = single_conv(x, p, f"{pre}.basic_module.SingleConv1", num_groups)
    return single_conv(x, p, f"{pre}.basic_module.SingleConv2", num_groups)


def unet3d_forward(p, x, num_levels=4, num_groups=8, upsample="default"):
    feats = []
    for i in range(num_levels):
        if i > 0:
            x = F.max_pool3d(x, 2)
        x = double_conv(x, p, f"encoders.{i}", num_groups)
        feats.insert(0, x)
    feats = feats[1:]
    for i, enc in enumerate(feats):
        if upsample == "deconv":   # TransposeConvUpsampling (buildingblocks.py:676-728): ConvTranspose3d(k3, s2, p1, no bias), then resize
            x = F.conv_transpose3d(x, p[f"decoders.{i}.upsampling.upsample.conv_transposed.weight"], None, stride=2, padding=1)
        x = F.interpolate(x, size=enc.shape[2:], mode="nearest")
        x = torch.cat((enc, x), dim=1)
        x = double_conv(x, p, f"decoders.{i}", num_groups)
    return F.conv3d(x, p["final_conv.weight"], p["final_conv.bias"])


def flatten(t):
    C = t.size(1)
    order = (1, 0) + tuple(range(2, t.dim()))
    return t.permute(order).contiguous().view(C, -1)


def compute_per_channel_dice(inp, target, epsilon=1e-6):
    inp = flatten(inp)
    target = flatten(target).float()
    intersect = (inp * target).sum(-1)
    denom = (inp * inp).sum(-1) + (target * target).sum(-1)
    return 2 * (intersect / denom.clamp(min=epsilon))


def dice_loss(logits, target):
    return 1.0 - torch.mean(compute_per_channel_dice(torch.sigmoid(logits), target))


def bce_dice_loss(logits, target, alpha=1.0, beta=1.0):
    return alpha * F.binary_cross_entropy_with_logits(logits, target) + beta * dice_loss(logits, target)


def hf_wrapper_loss(logits, target, alpha=1.0, beta=1.0):
    """model/unet3d/UNet3D.py:134-154: activation first, then BCEDice on the activated output."""
    return bce_dice_loss(torch.sigmoid(logits), target, alpha, beta)


def loss_and_grads(p, x, target, num_levels=4, num_groups=8, upsample="default"):
    ps = {k: v.detach().clone().requires_grad_(True) for k, v in p.items()}
    logits = unet3d_forward(ps, x, num_levels, num_groups, upsample)
    loss = bce_dice_loss(logits, target)
    loss.backward()
    return loss.detach(), logits.detach(), {k: v.grad.detach() for k, v in ps.items()}


# ---- bf16-STORAGE emulation (test infrastructure for the 3-D engine's bf16 mode; the 2-D twin is unet2d_oracle.unet_forward_bf16_storage) ----------------
# The engine's bf16 mode keeps in bf16: every activation (the outputs of ReLU / max-pool), the NORMALISED operand of each convolution (mis_gn_apply writes
# round(scale * x + shift); GroupNorm statistics are taken from the stored bf16 input in fp32), the packed conv weights and every activation gradient;
# in fp32: the volume, the first layer's and the head's weights, the accumulators, every parameter gradient and the master weights.  Restated below on top of
# the fp32 oracle with round-to-nearest-even at the same tensor boundaries.
def unet3d_forward_bf16_storage(p, x, num_levels=4, num_groups=8):
    from .unet2d_oracle import _RoundAct, _RoundWeight
    ra, rw = _RoundAct.apply, _RoundWeight.apply

    def sc(t, pre, first=False):
        g = _groups(t.shape[1], num_groups)
        tn = F.group_norm(t, g, p[f"{pre}.groupnorm.weight"], p[f"{pre}.groupnorm.bias"], eps=1e-5)
        w = p[f"{pre}.conv.weight"]
        if not first:
            tn, w = ra(tn), rw(w)
        return ra(F.relu(F.conv3d(tn, w, None, padding=1)))

    def dc(t, pre, first=False):
        return sc(sc(t, f"{pre}.basic_module.SingleConv1", first), f"{pre}.basic_module.SingleConv2")

    feats = []
    for i in range(num_levels):
        if i > 0:
            x = F.max_pool3d(x, 2)
        x = dc(x, f"encoders.{i}", first=(i == 0))
        feats.insert(0, x)
    feats = feats[1:]
    for i, enc in enumerate(feats):
        x = F.interpolate(x, size=enc.shape[2:], mode="nearest")
        x = dc(torch.cat((enc, x), dim=1), f"decoders.{i}")
    return F.conv3d(x, p["final_conv.weight"], p["final_conv.bias"])


def loss_and_grads_bf16_storage(p, x, target, num_levels=4, num_groups=8):
    ps = {k: v.detach().clone().requires_grad_(True) for k, v in p.items()}
    logits = unet3d_forward_bf16_storage(ps, x, num_levels, num_groups)
    loss = bce_dice_loss(logits, target)
    loss.backward()
    return loss.detach(), logits.detach(), {k: v.grad.detach() for k, v in ps.items()}


def _res_block(x, p, pre, num_groups=8):
    """ResNetBlock with order 'gcr' (buildingblocks.py:255-325): optional 1x1x1 conv, SingleConv 'gcr', SingleConv 'gc', += residual, ReLU"""
    r = F.conv3d(x, p[f"{pre}.conv1.weight"], p[f"{pre}.conv1.bias"]) if f"{pre}.conv1.weight" in p else x
    g = _groups(r.shape[1], num_groups)
    t = F.group_norm(r, g, p[f"{pre}.conv2.groupnorm.weight"], p[f"{pre}.conv2.groupnorm.bias"], eps=1e-5)
    t = F.relu(F.conv3d(t, p[f"{pre}.conv2.conv.weight"], None, padding=1))
    u = F.group_norm(t, g, p[f"{pre}.conv3.groupnorm.weight"], p[f"{pre}.conv3.groupnorm.bias"], eps=1e-5)
    u = F.conv3d(u, p[f"{pre}.conv3.conv.weight"], None, padding=1)
    out = F.relu(u + r)
    m = f"{pre}.se_module"
    if f"{m}.cSE.fc1.weight" in p:          # ResNetBlockSE, se_module 'scse' (buildingblocks.py:326-362; se.py:18-116, reduction_ratio 1)
        mean = out.mean(dim=(2, 3, 4))
        a = torch.sigmoid(F.linear(F.relu(F.linear(mean, p[f"{m}.cSE.fc1.weight"], p[f"{m}.cSE.fc1.bias"])), p[f"{m}.cSE.fc2.weight"], p[f"{m}.cSE.fc2.bias"]))
        b = torch.sigmoid(F.conv3d(out, p[f"{m}.sSE.conv.weight"], p[f"{m}.sSE.conv.bias"]))
        out = torch.max(out * a[:, :, None, None, None], out * b)
    return out


def resunet3d_forward(p, x, num_levels, num_groups=8):
    """ResidualUNet3D (model.py:197-232): ResNetBlock encoders with MaxPool3d, decoders = ConvTranspose3d(k3, s2, p1) resized to the encoder grid,
    SUM joining, ResNetBlock; 1x1x1 head.  With `se_module` parameters in p: ResidualUNetSE3D (model.py:235-280)."""
    feats = []
    for i in range(num_levels):
        if i > 0:
            x = F.max_pool3d(x, 2)
        x = _res_block(x, p, f"encoders.{i}.basic_module", num_groups)
        feats.insert(0, x)
    feats = feats[1:]
    for i, enc in enumerate(feats):
        x = F.conv_transpose3d(x, p[f"decoders.{i}.upsampling.upsample.conv_transposed.weight"], None, stride=2, padding=1)
        x = F.interpolate(x, size=enc.shape[2:], mode="nearest")
        x = _res_block(enc + x, p, f"decoders.{i}.basic_module", num_groups)
    return F.conv3d(x, p["final_conv.weight"], p["final_conv.bias"])
