"""CPU restatement (PyTorch) of the reference's squeeze & excitation layers, model/unet3d/se.py.  TEST INFRASTRUCTURE ONLY: tests/ compare the HIP layers
(csrc/se3d.hip through mdeical_image_segmentation_amd.blocks3d.se) with these functions; nothing in the product imports this file.

Pinned by tests/golden/g20_se_layers.npz (outputs and gradients of the reference's own classes, tests/golden/make_golden_se_layers.py)."""
import torch
import torch.nn.functional as F


def cse(x, fc1_w, fc1_b, fc2_w, fc2_b):
    """ChannelSELayer3D.forward (se.py:40-53): gate = sigmoid(fc2(relu(fc1(mean over D, H, W)))) per (sample, channel)"""
    m = x.mean(dim=(2, 3, 4))                                            # AdaptiveAvgPool3d(1) (:43)
    a = torch.sigmoid(F.linear(F.relu(F.linear(m, fc1_w, fc1_b)), fc2_w, fc2_b))      # (:46-47)
    return x * a[:, :, None, None, None]                                 # (:49)


def sse(x, conv_w, conv_b):
    """SpatialSELayer3D.forward (se.py:72-98): gate = sigmoid(1x1x1 conv to one channel) per voxel"""
    return x * torch.sigmoid(F.conv3d(x, conv_w, conv_b))                # (:89-96)


def scse(x, fc1_w, fc1_b, fc2_w, fc2_b, conv_w, conv_b):
    """ChannelSpatialSELayer3D.forward (se.py:114-116): elementwise max of the two"""
    return torch.max(cse(x, fc1_w, fc1_b, fc2_w, fc2_b), sse(x, conv_w, conv_b))
