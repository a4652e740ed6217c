"""CPU restatement (PyTorch) of the reference's SegmentationLoss, model/unet2d/loss.py:21-70.  TEST INFRASTRUCTURE ONLY.

PARITY UNPINNED for the MS-SSIM term: it lives in the third-party package pytorch_msssim (pinned 1.0.0 in the reference's requirements.txt:127),
which is absent from this container and from /root/reference; its published algorithm (ssim.py: `_fspecial_gauss_1d`, `gaussian_filter`, `_ssim`,
`ms_ssim`) is restated below.  F1Loss / IoULoss are the reference's own lines (:32-56)."""
import torch
import torch.nn.functional as F

WEIGHTS = [0.0448, 0.2856, 0.3001, 0.2363, 0.1333]


def _window(size=11, sigma=1.5):
    coords = torch.arange(size, dtype=torch.float)
    coords -= size // 2
    g = torch.exp(-(coords ** 2) / (2 * sigma ** 2))
    return (g / g.sum()).view(1, 1, 1, -1)


def _gauss(x, win):      # separable "valid" filtering, H then W, groups = channels (1)
    x = F.conv2d(x, win.transpose(2, 3), padding=0)
    return F.conv2d(x, win, padding=0)


def _ssim(X, Y, win, data_range=1.0, K=(0.01, 0.03)):
    C1, C2 = (K[0] * data_range) ** 2, (K[1] * data_range) ** 2
    mu1, mu2 = _gauss(X, win), _gauss(Y, win)
    mu1_sq, mu2_sq, mu1_mu2 = mu1.pow(2), mu2.pow(2), mu1 * mu2
    s1 = _gauss(X * X, win) - mu1_sq
    s2 = _gauss(Y * Y, win) - mu2_sq
    s12 = _gauss(X * Y, win) - mu1_mu2
    cs_map = (2 * s12 + C2) / (s1 + s2 + C2)
    ssim_map = ((2 * mu1_mu2 + C1) / (mu1_sq + mu2_sq + C1)) * cs_map
    return torch.flatten(ssim_map, 2).mean(-1), torch.flatten(cs_map, 2).mean(-1)


def ms_ssim(X, Y):
    win = _window().to(X.dtype)
    w = X.new_tensor(WEIGHTS)
    mcs = []
    for i in range(5):
        ssim_pc, cs = _ssim(X, Y, win)
        if i < 4:
            mcs.append(torch.relu(cs))
            pad = [s % 2 for s in X.shape[2:]]
            X = F.avg_pool2d(X, kernel_size=2, padding=pad)
            Y = F.avg_pool2d(Y, kernel_size=2, padding=pad)
    ssim_pc = torch.relu(ssim_pc)
    stack = torch.stack(mcs + [ssim_pc], dim=0)
    return torch.prod(stack ** w.view(-1, 1, 1), dim=0).mean()


def msssim_loss(inputs, targets):
    return 1 - ms_ssim(torch.sigmoid(inputs), targets)


def iou_loss(inputs, targets, eps=1e-7):
    p = torch.sigmoid(inputs)
    inter = (p * targets).sum()
    union = p.sum() + targets.sum() - inter
    return 1 - (inter + eps) / (union + eps)


def f1_loss(inputs, targets, eps=1e-7):
    p = torch.sigmoid(inputs)
    TP = (p * targets).sum()
    precision = TP / (p.sum() + eps)
    recall = TP / (targets.sum() + eps)
    return 1 - 2 * (precision * recall) / (precision + recall + eps)


def segmentation_loss(inputs, targets):
    return f1_loss(inputs, targets) + msssim_loss(inputs, targets) + iou_loss(inputs, targets)
