"""ORACLE (test infrastructure, not product code) - numpy restatement of the reference's 3-D augmentation
(augment/unet3d_augment/transforms.py): RandomFlip :25-50, RandomRotate90 :53-80, RandomRotate :83-112 with order 0
(scipy.ndimage.rotate restated from its published algorithm: per-plane affine transform, coordinate =
((0 + i*m[d][0]) + j*m[d][1]) + offset[d], reflect the coordinate about the half-sample edges, round half up, reflect
the index), RandomContrast :115-133, Standardize :495-523.  Pinned against tests/golden/g5_augment.npz, generated from
the real reference classes (which call scipy itself)."""
import numpy as np
from scipy import special


def flip(m, mask):
    for axis in (0, 1, 2):
        if mask & (1 << axis):
            m = np.flip(m, axis)
    return m


def rot90(m, k):
    return np.rot90(m, k, (1, 2))


def _refl_coord(x, n):
    x = x.copy()
    sz2 = 2 * n
    neg = x < 0
    far = neg & (x < -sz2)
    x[far] = sz2 * np.trunc(-x[far] / sz2) + x[far]
    lo = neg & (x < -n)
    mid = neg & ~lo
    x[lo] = x[lo] + sz2
    xm = x[mid]
    x[mid] = np.where(xm > -1e-15, 1e-15, -xm) - 1.0
    pos = (~neg) & (x > n - 1)
    xp = x[pos] - sz2 * np.trunc(x[pos] / sz2)
    xp = np.where(xp >= n, sz2 - xp - 1, xp)
    x[pos] = xp
    return x


def _refl_idx(k, n):
    k = k.copy()
    sz2 = 2 * n
    neg = k < 0
    far = neg & (k < -sz2)
    k[far] = sz2 * (-k[far] // sz2) + k[far]
    lo = neg & (k < -n)
    mid = neg & ~lo
    k[lo] += sz2
    k[mid] = -k[mid] - 1
    pos = (~neg) & (k > n - 1)
    kp = k[pos] - sz2 * (k[pos] // sz2)
    kp = np.where(kp >= n, sz2 - kp - 1, kp)
    k[pos] = kp
    return k


def rotate0(m, angle, axes):
    """scipy.ndimage.rotate(m, angle, axes=axes, reshape=False, order=0, mode='reflect') for a 3-D array."""
    a0, a1 = sorted(axes)
    c, s = special.cosdg(angle), special.sindg(angle)
    rot = np.array([[c, s], [-s, c]])
    shp = np.asarray(m.shape)[[a0, a1]]
    off = (shp - 1) / 2 - rot @ ((shp - 1) / 2)
    n0, n1 = int(shp[0]), int(shp[1])
    i = np.arange(n0, dtype=np.float64)[:, None]
    j = np.arange(n1, dtype=np.float64)[None, :]
    x0 = ((0.0 + i * rot[0, 0]) + j * rot[0, 1]) + off[0]
    x1 = ((0.0 + i * rot[1, 0]) + j * rot[1, 1]) + off[1]
    k0 = _refl_idx(np.floor(_refl_coord(x0, n0) + 0.5).astype(np.int64), n0)
    k1 = _refl_idx(np.floor(_refl_coord(x1, n1) + 0.5).astype(np.int64), n1)
    mm = np.moveaxis(m, (a0, a1), (0, 1))
    out = mm[k0, k1]
    return np.moveaxis(out, (0, 1), (a0, a1))


def contrast(m, mean, alpha):
    return np.clip(mean + alpha * (m - mean), -1, 1)


def standardize(m, mean=None, std=None, eps=1e-10):
    if mean is None:
        mean, std = np.mean(m), np.std(m)
    return (m - mean) / np.clip(std, a_min=eps, a_max=None)


class Pipeline:
    """Replays the reference's draw order (one RandomState(seed) per transform) for
    raw:   RandomFlip, RandomRotate90, RandomRotate(axes, order 0), RandomContrast(p)
    label: RandomFlip, RandomRotate90, RandomRotate(axes, order 0)"""

    def __init__(self, seed, axes, contrast_p, mean=0.0, alpha=(0.5, 1.5), spectrum=30):
        self.rs = {k: np.random.RandomState(seed) for k in ("rflip", "rrot90", "rrot", "rcon", "lflip", "lrot90", "lrot")}
        self.axes, self.p, self.mean, self.alpha, self.spectrum = axes, contrast_p, mean, alpha, spectrum

    def _geo(self, m, f, r90, rr):
        mask = 0
        for axis in (0, 1, 2):
            if f.uniform() > 0.5:
                mask |= 1 << axis
        m = flip(m, mask)
        m = rot90(m, r90.randint(0, 4))
        axis = self.axes[rr.randint(len(self.axes))]
        angle = rr.randint(-self.spectrum, self.spectrum)
        return rotate0(m, angle, axis)

    def raw(self, m):
        m = self._geo(m, self.rs["rflip"], self.rs["rrot90"], self.rs["rrot"])
        if self.rs["rcon"].uniform() < self.p:
            a = self.rs["rcon"].uniform(self.alpha[0], self.alpha[1])
            m = contrast(m, self.mean, a)
        return m

    def label(self, m):
        return self._geo(m, self.rs["lflip"], self.rs["lrot90"], self.rs["lrot"])
