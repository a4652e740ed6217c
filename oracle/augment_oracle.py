"""ORACLE (test infrastructure, not product code) - numpy restatement of the reference's 3-D augmentation
(augment/unet3d_augment/transforms.py): RandomFlip :25-50, RandomRotate90 :53-80, RandomRotate :83-112 with order 0
(scipy.ndimage.rotate restated from its published algorithm: per-plane affine transform, coordinate =
((0 + i*m[d][0]) + j*m[d][1]) + offset[d], reflect the coordinate about the half-sample edges, round half up, reflect
the index), RandomContrast :115-133, Standardize :495-523.  Pinned against tests/golden/g5_augment.npz, generated from
the real reference classes (which call scipy itself)."""
import math

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


_POLE3 = np.sqrt(3.0) - 2.0


def _map_coord_mode(x, n, mode):
    """scipy ni_interpolation.c map_coordinate on a vector of double coordinates (the boundary modes of scipy.ndimage.rotate / affine_transform)"""
    x = x.copy()
    if mode == "grid-constant":
        return x
    neg, pos = x < 0, x > n - 1
    if n <= 1 and mode in ("mirror", "reflect", "grid-mirror", "wrap", "grid-wrap"):
        x[neg | pos] = 0
        return x
    if mode == "mirror":
        sz2 = 2 * n - 2
        t = sz2 * np.trunc(-x[neg] / sz2) + x[neg]
        x[neg] = np.where(t <= 1 - n, t + sz2, -t)
        t = x[pos] - sz2 * np.trunc(x[pos] / sz2)
        x[pos] = np.where(t >= n, sz2 - t, t)
    elif mode in ("reflect", "grid-mirror"):
        sz2 = 2 * n
        t = x[neg]
        t = np.where(t < -sz2, sz2 * np.trunc(-t / sz2) + t, t)
        x[neg] = np.where(t < -n, t + sz2, np.where(t > -1e-15, 1e-15, -t) - 1)
        t = x[pos] - sz2 * np.trunc(x[pos] / sz2)
        x[pos] = np.where(t >= n, sz2 - t - 1, t)
    elif mode == "wrap":
        sz = n - 1
        x[neg] = x[neg] + sz * (np.trunc(-x[neg] / sz) + 1)
        x[pos] = x[pos] - sz * np.trunc(x[pos] / sz)
    elif mode == "grid-wrap":
        x[neg] = x[neg] + n * (np.trunc((-1 - x[neg]) / n) + 1)
        x[pos] = x[pos] - n * np.trunc(x[pos] / n)
    elif mode == "nearest":
        x[neg] = 0
        x[pos] = n - 1
    elif mode == "constant":
        x[neg | pos] = -1
    else:
        raise ValueError(mode)
    return x


def _idx_mode(i, n, mode):
    """a rounded support index outside [0, n): the sample scipy reads instead, and whether the output is cval"""
    inside = np.zeros(i.shape, bool)
    if mode in ("reflect", "grid-mirror"):
        m = np.mod(i, 2 * n)
        return np.where(m < n, m, 2 * n - 1 - m), inside
    if mode == "mirror":
        if n == 1:
            return np.zeros_like(i), inside
        m = np.mod(i, 2 * n - 2)
        return np.where(m < n, m, 2 * n - 2 - m), inside
    if mode in ("wrap", "grid-wrap"):
        return np.mod(i, n), inside
    if mode == "nearest":
        return np.clip(i, 0, n - 1), inside
    return np.clip(i, 0, n - 1), (i < 0) | (i >= n)          # constant / grid-constant


def rotate0_modes(m, angle, axes, mode="reflect", cval=-1):
    """scipy.ndimage.rotate(m, angle, axes, reshape=False, order=0, mode=mode, cval=cval) for EVERY boundary mode (reference transforms.py:109-111 passes the
    constructor's mode through): bit-identical to scipy (tests/test_augment.py); mis_aug_rotate0_mode (csrc/augment.hip) is the device form"""
    a0, a1 = sorted(axes)
    c, s = special.cosdg(angle), special.sindg(angle)
    rot = np.array([[c, s], [-s, c]])
    shp = np.asarray(m.shape)[[a0, a1]]
    off = (shp - 1) / 2 - rot @ ((shp - 1) / 2)
    idx = np.indices(m.shape)
    o0, o1 = idx[a0].astype(np.float64), idx[a1].astype(np.float64)
    x0 = ((0.0 + o0 * rot[0, 0]) + o1 * rot[0, 1]) + off[0]
    x1 = ((0.0 + o0 * rot[1, 0]) + o1 * rot[1, 1]) + off[1]
    n0, n1 = m.shape[a0], m.shape[a1]
    x0m, x1m = _map_coord_mode(x0.ravel(), n0, mode), _map_coord_mode(x1.ravel(), n1, mode)
    const = (x0m <= -1.0) | (x1m <= -1.0) if mode == "constant" else np.zeros(x0m.shape, bool)
    i0, c0 = _idx_mode(np.floor(x0m + 0.5).astype(np.int64), n0, mode)
    i1, c1 = _idx_mode(np.floor(x1m + 0.5).astype(np.int64), n1, mode)
    ii = [idx[k].ravel() for k in range(m.ndim)]
    ii[a0], ii[a1] = i0, i1
    out = np.where(const | c0 | c1, np.asarray(cval, m.dtype), m[tuple(ii)])
    return out.reshape(m.shape)


def spline_filter3_line(c):
    """scipy ni_splines.c (scipy 1.15, third-party): cubic B-spline prefilter of one line, mode 'reflect' (half-sample symmetric):
    gain (1-z)(1-1/z), exact causal initialisation over the reflected signal, causal and anti-causal recursions.
    c: float64 array, filtered along axis 0 (vectorised over the other axes)."""
    z = _POLE3
    n = c.shape[0]
    c = c * ((1.0 - z) * (1.0 - 1.0 / z))
    if n == 1:
        return c
    z_n = z ** n
    c0 = c[0].copy()
    acc = c[0] + z_n * c[n - 1]
    z_i = z
    for i in range(1, n):
        acc = acc + z_i * (c[i] + z_n * c[n - 1 - i])
        z_i *= z
    acc = acc * (z / (1.0 - z_n * z_n))
    c[0] = acc + c0
    for i in range(1, n):
        c[i] = c[i] + z * c[i - 1]
    c[n - 1] = c[n - 1] * (z / (z - 1.0))
    for i in range(n - 2, -1, -1):
        c[i] = z * (c[i + 1] - c[i])
    return c


def rotate3(m, angle, axes):
    """scipy.ndimage.rotate(m, angle, axes=axes, reshape=False, order=3, mode='reflect') for a 3-D float array: per plane,
    spline_filter (float64, axis 0 then axis 1 of the plane) and cubic interpolation at the rotated coordinates; support points
    beyond the edges follow the 'reflect' index rule; result cast to m.dtype."""
    a0, a1 = sorted(axes)
    c, s = special.cosdg(angle), special.sindg(angle)
    rot = np.array([[c, s], [-s, c]])
    shp = np.asarray(m.shape)[[a0, a1]]
    off = (shp - 1) / 2 - rot @ ((shp - 1) / 2)
    n0, n1 = int(shp[0]), int(shp[1])
    mm = np.moveaxis(m, (a0, a1), (0, 1)).astype(np.float64)
    co = spline_filter3_line(mm.copy())
    co = np.moveaxis(spline_filter3_line(np.moveaxis(co, 1, 0).copy()), 0, 1)
    i = np.arange(n0, dtype=np.float64)[:, None]
    j = np.arange(n1, dtype=np.float64)[None, :]
    x0 = _refl_coord(((0.0 + i * rot[0, 0]) + j * rot[0, 1]) + off[0], n0)
    x1 = _refl_coord(((0.0 + i * rot[1, 0]) + j * rot[1, 1]) + off[1], n1)

    def weights(x):
        f = np.floor(x)
        y = x - f
        zc = 1.0 - y
        w1 = (y * y * (y - 2.0) * 3.0 + 4.0) / 6.0
        w2 = (zc * zc * (zc - 2.0) * 3.0 + 4.0) / 6.0
        w0 = zc * zc * zc / 6.0
        w3 = 1.0 - w0 - w1 - w2
        return f.astype(np.int64) - 1, (w0, w1, w2, w3)

    s0, w0 = weights(x0)
    s1, w1 = weights(x1)
    out = np.zeros((n0, n1) + mm.shape[2:], dtype=np.float64)
    for a in range(4):
        k0 = _refl_idx(s0 + a, n0)
        for b in range(4):
            k1 = _refl_idx(s1 + b, n1)
            wgt = (w0[a] * w1[b]).reshape((n0, n1) + (1,) * (mm.ndim - 2))
            cf = co[k0, k1]
            out = out + (cf * w0[a].reshape(wgt.shape)) * w1[b].reshape(wgt.shape)
    return np.moveaxis(out, (0, 1), (a0, a1)).astype(m.dtype)


def elastic(m, rs, spline_order, alpha=2000, sigma=50, apply_3d=True):
    """ElasticDeformation.__call__ after the execution-probability draw (transforms.py:167-189): the smoothing and the resampling ARE scipy
    calls in the reference (scipy is present in this container, so the oracle calls the same functions); draws randn fields from `rs`."""
    from scipy.ndimage import gaussian_filter, map_coordinates
    shape = m.shape if m.ndim == 3 else m[0].shape
    dz = gaussian_filter(rs.randn(*shape), sigma, mode="reflect") * alpha if apply_3d else np.zeros(shape)
    dy, dx = [gaussian_filter(rs.randn(*shape), sigma, mode="reflect") * alpha for _ in range(2)]
    z, y, x = np.meshgrid(np.arange(shape[0]), np.arange(shape[1]), np.arange(shape[2]), indexing="ij")
    idx = z + dz, y + dy, x + dx
    if m.ndim == 3:
        return map_coordinates(m, idx, order=spline_order, mode="reflect")
    return np.stack([map_coordinates(c, idx, order=spline_order, mode="reflect") for c in m], axis=0)


def gaussian_blur3d(m, sigma):
    """GaussianBlur3D.__call__ after its two `random` draws (transforms.py:713-718): `skimage.filters.gaussian(x, sigma=sigma)`.  scikit-image (pinned 0.23.2 in
    the reference's requirements.txt:143) is absent from this image; its published algorithm for a float image without channel_axis
    (skimage/filters/_gaussian.py: mode='nearest', cval=0, truncate=4.0, preserve_range=False leaves float data unscaled, float32 stays float32) is
    exactly `scipy.ndimage.gaussian_filter(image, sigma, output=<same float dtype>, mode='nearest', cval=0, truncate=4.0)`, and scipy is present here."""
    from scipy.ndimage import gaussian_filter
    m = np.asarray(m)
    ft = np.float32 if m.dtype in (np.float16, np.float32) else np.float64
    return gaussian_filter(m.astype(ft), sigma, mode="nearest", cval=0, truncate=4.0)


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


def crop_to_fixed(m, rs, size, centered=False):
    """CropToFixed (transforms.py:194-247) as index arithmetic: window start from the stream (a draw even when the range is 1), then
    out[..., i, j] = m[..., r(y0 - pad_lo_y + i), r(x0 - pad_lo_x + j)] with numpy's 'reflect' rule r (period 2n-2)."""
    cy, cx = size
    y, x = m.shape[-2:]

    def start_pad(crop, mx):
        if crop < mx:
            return (mx - crop) // 2 if centered else None, 0
        return 0, (crop - mx) // 2

    (ys, ylo), (xs, xlo) = start_pad(cy, y), start_pad(cx, x)
    if not centered:
        ys = rs.randint(y - cy if cy < y else 1)
        xs = rs.randint(x - cx if cx < x else 1)

    def refl(idx, n):
        if n == 1:
            return np.zeros_like(idx)
        p = 2 * n - 2
        idx = np.mod(idx, p)
        return np.where(idx < n, idx, p - idx)

    iy = refl(ys - ylo + np.arange(cy), y)
    ix = refl(xs - xlo + np.arange(cx), x)
    return m[..., iy[:, None], ix[None, :]]


def poisson_noise(m, rs, lam_range):
    """AdditivePoissonNoise (transforms.py:622-633) after the execution draw: lam = uniform(range); m + poisson(lam) (float64 like numpy's promotion)."""
    lam = rs.uniform(lam_range[0], lam_range[1])
    return m + rs.poisson(lam, size=m.shape)


# ---- AdditiveGaussianNoise on the reference's stream (transforms.py:608-619) ------------------------------------------------------------------
# The field comes from numpy's LEGACY RandomState (third-party: numpy, reference pin numpy==1.26.4, requirements.txt; algorithm unchanged since 1.17):
# numpy/random/src/mt19937/mt19937.c (mt19937_gen / tempering), numpy/random/src/legacy/legacy-distributions.c (legacy_double, legacy_gauss, legacy_normal).
# Restated here word by word - it is the checker of csrc/mt19937.hip and is itself pinned against a golden drawn by the real reference class.
def mt19937_words(key, pos, n):
    """the next n 32-bit outputs of MT19937 from (key[624] uint32, pos); returns (words, key', pos')"""
    mt = np.array(key, dtype=np.uint64).copy()
    out = np.empty(n, dtype=np.uint32)
    done = 0
    U, L, A = np.uint64(0x80000000), np.uint64(0x7fffffff), np.uint64(0x9908b0df)
    while done < n:
        if pos >= 624:
            for kk in range(624):                       # mt19937_gen: in place, in index order
                y = (mt[kk] & U) | (mt[(kk + 1) % 624] & L)
                mt[kk] = mt[(kk + 397) % 624] ^ (y >> np.uint64(1)) ^ (A if (y & np.uint64(1)) else np.uint64(0))
            pos = 0
        take = min(n - done, 624 - pos)
        y = mt[pos:pos + take].copy()
        y ^= (y >> np.uint64(11))
        y ^= (y << np.uint64(7)) & np.uint64(0x9d2c5680)
        y ^= (y << np.uint64(15)) & np.uint64(0xefc60000)
        y ^= (y >> np.uint64(18))
        out[done:done + take] = (y & np.uint64(0xffffffff)).astype(np.uint32)
        done += take
        pos += take
    return out, (mt & np.uint64(0xffffffff)).astype(np.uint32), pos


def legacy_normal_field(words, count, scale, has_gauss=0, cached=0.0):
    """count samples of loc=0, scale*legacy_gauss drawn from the 32-bit word stream; returns (samples float64, words consumed, has_gauss', cached')"""
    out = np.empty(count, dtype=np.float64)
    w = 0
    i = 0

    def dbl():
        nonlocal w
        a, b = int(words[w]) >> 5, int(words[w + 1]) >> 6
        w += 2
        return (a * 67108864.0 + b) / 9007199254740992.0

    while i < count:
        if has_gauss:
            g, has_gauss, cached = cached, 0, 0.0
        else:
            while True:
                x1 = 2.0 * dbl() - 1.0
                x2 = 2.0 * dbl() - 1.0
                r2 = x1 * x1 + x2 * x2
                if r2 < 1.0 and r2 != 0.0:
                    break
            f = math.sqrt(-2.0 * math.log(r2) / r2)
            cached, has_gauss = f * x1, 1
            g = f * x2
        out[i] = 0.0 + scale * g
        i += 1
    return out, w, has_gauss, cached


def additive_gaussian_noise(m, rs, scale_range, execution_probability):
    """the whole transform on a numpy RandomState `rs` WITHOUT calling rs.normal: decision and std from the stream as the reference draws them, the field from the
    restatement above; the generator is left where the reference leaves it."""
    if not (rs.uniform() < execution_probability):
        return m
    std = rs.uniform(scale_range[0], scale_range[1])
    name, key, pos, has_gauss, cached = rs.get_state()
    n = m.size
    need = int((n * 1.3 + 200)) * 2 * 2
    words, _, _ = mt19937_words(key, pos, need)
    field, used, hg, cv = legacy_normal_field(words, n, std, has_gauss, cached)
    if used:
        rs.randint(0, 4294967296, size=used, dtype=np.uint32)
    st = rs.get_state()
    rs.set_state((st[0], st[1], st[2], hg, cv))
    return m.astype(np.float64) + field.reshape(m.shape)
