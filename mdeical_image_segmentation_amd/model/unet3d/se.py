"""Mirror of the reference's model/unet3d/se.py (ChannelSELayer3D :18-53, SpatialSELayer3D :56-98, ChannelSpatialSELayer3D :101-116): the same module tree,
state-dict keys (`cSE.fc1.{weight,bias}`, `cSE.fc2.{weight,bias}`, `sSE.conv.{weight,bias}`), PyTorch default initialisation in the reference's construction order.

Two routes: inside ResidualUNetSE3D the fused residual engine (engine3d_res.ResidualUNetSE3DEngine) reads the parameters and runs csrc/se3d.hip fused with the block's
ReLU; called ON ITS OWN, `layer(x)` with x = (N, C, D, H, W) on the GPU runs the same kernels through `blocks3d.se` (mis_se_layer_*: one statistics pass, the two small
fully connected layers per sample, one gate-and-multiply pass; backward = one reducing pass, the fc gradients, one apply pass - all HIP, no CPU path)."""
from torch import nn

from ... import blocks3d as B
from ... import ops


class _SELayer(nn.Module):
    def forward(self, x):
        return B.from_cl(self._cl(B.to_cl(x)))


class ChannelSELayer3D(_SELayer):
    """se.py:18-53: x * sigmoid(fc2(relu(fc1(mean over D, H, W of x))))"""

    def __init__(self, num_channels, reduction_ratio=2):
        super().__init__()
        self.avg_pool = nn.AdaptiveAvgPool3d(1)            # (parameter-free members kept for the module tree; the arithmetic is the HIP route)
        self.reduction_ratio = reduction_ratio
        self.fc1 = nn.Linear(num_channels, num_channels // reduction_ratio, bias=True)
        self.fc2 = nn.Linear(num_channels // reduction_ratio, num_channels, bias=True)
        self.relu = nn.ReLU()
        self.sigmoid = nn.Sigmoid()

    def _cl(self, a):
        return B.se(a, ops.SE_CSE, cse=self)


class SpatialSELayer3D(_SELayer):
    """se.py:56-98: x * sigmoid(conv1x1x1(x) -> 1 channel).  The `weights` argument of the reference's forward (few-shot weights through F.conv2d on a 5-D tensor,
    :85-87) cannot run in the reference either and is refused."""

    def __init__(self, num_channels):
        super().__init__()
        self.conv = nn.Conv3d(num_channels, 1, 1)
        self.sigmoid = nn.Sigmoid()

    def forward(self, x, weights=None):
        if weights is not None:
            raise NotImplementedError("SpatialSELayer3D(weights=...): the reference applies F.conv2d to a 5-D tensor there (se.py:85-87), which raises; not built")
        return super().forward(x)

    def _cl(self, a):
        return B.se(a, ops.SE_SSE, sse=self)


class ChannelSpatialSELayer3D(_SELayer):
    """se.py:101-116: elementwise max of the two"""

    def __init__(self, num_channels, reduction_ratio=2):
        super().__init__()
        self.cSE = ChannelSELayer3D(num_channels, reduction_ratio)
        self.sSE = SpatialSELayer3D(num_channels)

    def _cl(self, a):
        return B.se(a, ops.SE_SCSE, cse=self.cSE, sse=self.sSE)
