"""CPU restatement (functional PyTorch) of the reference's UNet_3Plus forward, model/unet2d/unet.py:335-446, on a state-dict.
TEST INFRASTRUCTURE ONLY.  Pinned by tests/golden/g9_unet3plus.npz (the real module, train + eval mode)."""
import torch
import torch.nn.functional as F

FILTERS = [64, 128, 256, 512, 1024]


def _cbr(x, sd, conv, bn, training):
    x = F.conv2d(x, sd[conv + ".weight"], sd[conv + ".bias"], padding=1)
    rm, rv = sd[bn + ".running_mean"].clone(), sd[bn + ".running_var"].clone()
    x = F.batch_norm(x, rm, rv, sd[bn + ".weight"], sd[bn + ".bias"], training, 0.1, 1e-5)
    if training:
        sd[bn + ".running_mean"], sd[bn + ".running_var"] = rm, rv
    return F.relu(x)


def _unet_conv2(x, sd, pre, training):      # layers.py:8-46, two conv-bn-relu units named conv1 / conv2
    for i in (1, 2):
        x = _cbr(x, sd, f"{pre}.conv{i}.0", f"{pre}.conv{i}.1", training)
    return x


def branch_name(d, i):
    if i < d:
        return f"h{i}_PT_hd{d}"
    if i == d:
        return f"h{i}_Cat_hd{d}"
    return f"hd{i}_UT_hd{d}"


def forward(sd, x, training=True):
    """sd: state-dict (running statistics are REPLACED by their updated values in training mode)"""
    h = {1: _unet_conv2(x, sd, "conv1", training)}
    for i in range(2, 6):
        h[i] = _unet_conv2(F.max_pool2d(h[i - 1], 2), sd, f"conv{i}", training)                         # unet.py:337-349
    hd = {5: h[5]}
    for d in (4, 3, 2, 1):                                                                              # unet.py:352-443
        parts = []
        for i in range(1, 6):
            if i < d:
                k = 2 ** (d - i)
                src = F.max_pool2d(h[i], k, k, ceil_mode=True)
            elif i == d:
                src = h[i]
            else:
                src = F.interpolate(hd[i], scale_factor=2 ** (i - d), mode="bilinear")
            n = branch_name(d, i)
            parts.append(_cbr(src, sd, n + "_conv", n + "_bn", training))
        hd[d] = _cbr(torch.cat(parts, 1), sd, f"conv{d}d_1", f"bn{d}d_1", training)
    return F.conv2d(hd[1], sd["outconv1.weight"], sd["outconv1.bias"], padding=1)                     # unet.py:445-446
