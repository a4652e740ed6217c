"""Parameter containers with the module tree of the reference's model/unet3d/se.py (ChannelSELayer3D :18-53, SpatialSELayer3D :56-98,
ChannelSpatialSELayer3D :101-116): state-dict keys `cSE.fc1.{weight,bias}`, `cSE.fc2.{weight,bias}`, `sSE.conv.{weight,bias}` and PyTorch's default
initialisation in the reference's construction order.  The arithmetic runs in csrc/se3d.hip inside the fused residual engine
(engine3d_res.ResidualUNetSE3DEngine); calling a layer on its own raises."""
from torch import nn


class _ContainerOnly:
    def forward(self, *a, **k):
        raise NotImplementedError(f"{type(self).__name__} is a parameter container here: run the whole ResidualUNetSE3D (fused MI355X engine)")


class ChannelSELayer3D(_ContainerOnly, nn.Module):
    def __init__(self, num_channels, reduction_ratio=2):
        super().__init__()
        self.reduction_ratio = reduction_ratio
        self.fc1 = nn.Linear(num_channels, num_channels // reduction_ratio, bias=True)
        self.fc2 = nn.Linear(num_channels // reduction_ratio, num_channels, bias=True)


class SpatialSELayer3D(_ContainerOnly, nn.Module):
    def __init__(self, num_channels):
        super().__init__()
        self.conv = nn.Conv3d(num_channels, 1, 1)


class ChannelSpatialSELayer3D(_ContainerOnly, nn.Module):
    def __init__(self, num_channels, reduction_ratio=2):
        super().__init__()
        self.cSE = ChannelSELayer3D(num_channels, reduction_ratio)
        self.sSE = SpatialSELayer3D(num_channels)
