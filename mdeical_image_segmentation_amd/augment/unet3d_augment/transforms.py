"""On-device mirror of the reference's 3-D augmentation (augment/unet3d_augment/transforms.py).

Same class names, constructor arguments, seeding (`GLOBAL_RANDOM_STATE = RandomState(47)`, one seed per `Transformer`,
every transform gets its own `RandomState(seed)` - transforms.py:12,721-753) and the same draw order per call, so raw
and label pipelines stay in lock-step exactly like the reference.  The random PARAMETERS are drawn on the host from
numpy's MT19937 streams (the reference's own generator); the volumes never leave HBM: every transform is one HIP
gather / elementwise kernel (csrc/augment.hip).  Inputs may be numpy arrays (uploaded once) or CUDA tensors;
outputs are CUDA tensors.  Differences, all loud or documented:
  * RandomRotate supports order=0 (the class default, bit-exact vs scipy incl. 'reflect') and order=3 (cubic spline, fp64
    coefficients like scipy; equal to scipy up to float32 rounding); other orders raise;
  * AdditiveGaussianNoise draws the decision and the std from the reference stream but the noise FIELD comes from an
    on-device counter-based generator (same distribution, different values); `exact=True` generates the field with
    numpy on the host instead;
  * ElasticDeformation: spline_order 0 and 3 (fields from the reference's stream, smoothing + resampling on the device);
  * AdditivePoissonNoise: the integer field comes from the reference's stream on the host (uploaded, added on the device: bit-exact);
  * CropToFixed: window / mirror padding as one gather kernel, start positions from the reference's stream;
  * PercentileNormalizer: the order statistics are selected with torch.kthvalue on the device, numpy's interpolation rule;
  * label->boundary transforms (skimage find_boundaries / measure.label): out of scope.
"""
import ctypes as C
import importlib

import math

import numpy as np
import torch
from scipy import special

from ... import ops
from ..._lib import MisError, check, load, stream_ptr

GLOBAL_RANDOM_STATE = np.random.RandomState(47)


def _dev(m):
    if isinstance(m, np.ndarray):
        m = torch.from_numpy(np.ascontiguousarray(m)).cuda()
    if not isinstance(m, torch.Tensor) or m.device.type != "cuda":
        raise MisError("augment: expected a numpy array or a CUDA tensor (the transforms run on MI355X only)")
    if m.dim() not in (3, 4):
        raise AssertionError("Supports only 3D (DxHxW) or 4D (CxDxHxW) images")
    return m.contiguous()


def _flip_rot90(m, flipmask, k):
    m = _dev(m)
    if flipmask == 0 and (k & 3) == 0:
        return m
    D, H, W = m.shape[-3:]
    nvol = m.numel() // (D * H * W)
    es = m.element_size()
    if es not in (4, 8):
        raise MisError(f"augment: unsupported element size {es} ({m.dtype})")
    oshape = list(m.shape)
    if k & 1:
        oshape[-2], oshape[-1] = W, H
    out = torch.empty(oshape, dtype=m.dtype, device=m.device)
    check(load().mis_aug_flip_rot90(m.data_ptr(), out.data_ptr(), nvol, D, H, W, flipmask, k & 3, es, stream_ptr()), "mis_aug_flip_rot90")
    return out


class Compose(object):
    def __init__(self, transforms):
        self.transforms = transforms

    def __call__(self, m):
        for t in self.transforms:
            m = t(m)
        return m


class RandomFlip:
    """transforms.py:25-50: for axis in (0,1,2): flip if uniform() > axis_prob."""

    def __init__(self, random_state, axis_prob=0.5, **kwargs):
        assert random_state is not None, 'RandomState cannot be None'
        self.random_state = random_state
        self.axes = (0, 1, 2)
        self.axis_prob = axis_prob

    def __call__(self, m):
        mask = 0
        for axis in self.axes:
            if self.random_state.uniform() > self.axis_prob:
                mask |= 1 << axis
        return _flip_rot90(m, mask, 0)


class RandomRotate90:
    """transforms.py:53-80: k = randint(0, 4); np.rot90(m, k, axes=(1, 2)) (the H-W plane)."""

    def __init__(self, random_state, **kwargs):
        self.random_state = random_state
        self.axis = (1, 2)

    def __call__(self, m):
        k = self.random_state.randint(0, 4)
        return _flip_rot90(m, 0, int(k))


# scipy.ndimage boundary modes -> the mode codes of mis_aug_rotate0_mode (csrc/augment.hip)
_ROTATE_MODES = {"reflect": 0, "grid-mirror": 0, "constant": 1, "nearest": 2, "mirror": 3, "wrap": 4, "grid-wrap": 5, "grid-constant": 6}


class RandomRotate:
    """transforms.py:83-112: axis = axes[randint(len(axes))]; angle = randint(-spectrum, spectrum);
    scipy.ndimage.rotate(reshape=False, order, mode='reflect', cval=-1)."""

    def __init__(self, random_state, angle_spectrum=30, axes=None, mode='reflect', order=0, **kwargs):
        if axes is None:
            axes = [(1, 0), (2, 1), (2, 0)]
        else:
            assert isinstance(axes, list) and len(axes) > 0
        self.random_state = random_state
        self.angle_spectrum = angle_spectrum
        self.axes = axes
        self.mode = mode
        self.order = order

    def __call__(self, m):
        axis = self.axes[self.random_state.randint(len(self.axes))]
        angle = self.random_state.randint(-self.angle_spectrum, self.angle_spectrum)
        if self.order not in (0, 1, 2, 3, 4, 5) or self.mode not in _ROTATE_MODES or (self.order >= 1 and _ROTATE_MODES[self.mode] != 0):
            raise NotImplementedError("on-device RandomRotate: spline orders 0 .. 5 with mode='reflect' (the reference's default and the configs' setting), and every "
                                      "scipy boundary mode ('constant', 'nearest', 'mirror', 'wrap', 'grid-wrap', 'grid-constant', 'grid-mirror') at order 0, are built")
        m = _dev(m)
        if self.order >= 1 and m.dtype != torch.float32:
            raise MisError(f"RandomRotate(order={self.order}): fp32 volumes only (labels use order 0)")
        D, H, W = m.shape[-3:]
        a0, a1 = sorted(int(a) for a in axis)
        # exactly scipy.ndimage.rotate's host arithmetic
        c, s = special.cosdg(angle), special.sindg(angle)
        rot = np.array([[c, s], [-s, c]])
        shp = np.asarray((D, H, W))[[a0, a1]]
        out_center = rot @ ((shp - 1) / 2)
        in_center = (shp - 1) / 2
        off = in_center - out_center
        m4 = (C.c_double * 4)(rot[0, 0], rot[0, 1], rot[1, 0], rot[1, 1])
        o2 = (C.c_double * 2)(off[0], off[1])
        out = torch.empty_like(m)
        nvol = m.numel() // (D * H * W)
        if self.order >= 1:
            lib = load()
            ws = ops.workspace(lib.mis_aug_rotate3_workspace_bytes(nvol, D, H, W), m.device, "rotate3")
            if self.order == 3:
                check(lib.mis_aug_rotate3(m.data_ptr(), out.data_ptr(), ws.data_ptr(), nvol, D, H, W, a0, a1, m4, o2, stream_ptr()), "mis_aug_rotate3")
            else:
                check(lib.mis_aug_rotate_spline(m.data_ptr(), out.data_ptr(), ws.data_ptr(), nvol, D, H, W, a0, a1, m4, o2, int(self.order), stream_ptr()),
                      "mis_aug_rotate_spline")
            return out
        mode = _ROTATE_MODES[self.mode]
        if mode != 0:          # scipy.ndimage.rotate(..., mode=self.mode, cval=-1) (transforms.py:110): the fill value in the volume's own dtype
            cval = torch.tensor([-1], dtype=m.dtype).numpy().view(np.uint32 if m.element_size() == 4 else np.uint64)[0]
            check(load().mis_aug_rotate0_mode(m.data_ptr(), out.data_ptr(), nvol, D, H, W, a0, a1, m4, o2, m.element_size(), mode, int(cval), stream_ptr()),
                  "mis_aug_rotate0_mode")
            return out
        check(load().mis_aug_rotate0(m.data_ptr(), out.data_ptr(), nvol, D, H, W, a0, a1, m4, o2, m.element_size(), stream_ptr()),
              "mis_aug_rotate0")
        return out


class RandomContrast:
    """transforms.py:115-133: if uniform() < p: alpha = uniform(a0, a1); clip(mean + alpha * (m - mean), -1, 1)."""

    def __init__(self, random_state, alpha=(0.5, 1.5), mean=0.0, execution_probability=0.1, **kwargs):
        self.random_state = random_state
        assert len(alpha) == 2
        self.alpha = alpha
        self.mean = mean
        self.execution_probability = execution_probability

    def __call__(self, m):
        if self.random_state.uniform() < self.execution_probability:
            alpha = self.random_state.uniform(self.alpha[0], self.alpha[1])
            m = _dev(m)
            if m.dtype != torch.float32:
                raise MisError("RandomContrast: fp32 volumes only")
            out = torch.empty_like(m)
            check(load().mis_aug_contrast(m.data_ptr(), out.data_ptr(), m.numel(), float(self.mean), float(alpha), stream_ptr()), "mis_aug_contrast")
            return out
        return m


class ElasticDeformation:
    """transforms.py:138-191: with probability p, three (two when apply_3d=False) random fields randn(volume) are smoothed with
    scipy.ndimage.gaussian_filter(sigma, mode='reflect'), scaled by alpha and added to the voxel grid; the volume is resampled there with
    scipy.ndimage.map_coordinates(order=spline_order, mode='reflect').  The fields are drawn on the host from the reference's RandomState
    stream (same draw order) and uploaded as float64; smoothing (3 separable passes per field) and resampling run on the device
    (csrc/augment.hip).  spline_order 0 (labels; exact gather) and 3 (raw; float64 spline coefficients like scipy) are built."""

    def __init__(self, random_state, spline_order, alpha=2000, sigma=50, execution_probability=0.1, apply_3d=True, **kwargs):
        if spline_order not in (0, 3):
            raise NotImplementedError("on-device ElasticDeformation: spline_order 0 and 3 are built")
        self.random_state = random_state
        self.spline_order = spline_order
        self.alpha = alpha
        self.sigma = sigma
        self.execution_probability = execution_probability
        self.apply_3d = apply_3d

    def _field(self, shape, dev, wdev, radius):
        lib = load()
        a = torch.from_numpy(self.random_state.randn(*shape)).to(dev)                 # float64, the reference's stream
        b = torch.empty_like(a)
        D, H, W = shape
        for ax in range(3):                                                               # gaussian_filter: axis 0, 1, 2
            check(lib.mis_aug_gauss1d(a.data_ptr(), b.data_ptr(), 1, D, H, W, ax, wdev.data_ptr(), radius, stream_ptr()), "mis_aug_gauss1d")
            a, b = b, a
        return a

    def __call__(self, m):
        if self.random_state.uniform() < self.execution_probability:
            m = _dev(m)
            assert m.dim() in (3, 4)
            shape = tuple(m.shape[-3:])
            D, H, W = shape
            # scipy.ndimage.gaussian_filter1d: radius int(4*sigma + 0.5), normalised exp(-x^2 / (2 sigma^2)) in float64
            sd = float(self.sigma)
            radius = int(4.0 * sd + 0.5)
            xs = np.arange(-radius, radius + 1)
            phi = np.exp(-0.5 / (sd * sd) * xs ** 2)
            phi = (phi / phi.sum())[::-1].copy()
            wdev = torch.from_numpy(phi).to(m.device)
            fz = self._field(shape, m.device, wdev, radius) if self.apply_3d else None
            fy = self._field(shape, m.device, wdev, radius)
            fx = self._field(shape, m.device, wdev, radius)
            if self.spline_order == 3 and m.dtype != torch.float32:
                raise MisError("ElasticDeformation(spline_order=3): fp32 volumes only (labels use order 0)")
            out = torch.empty_like(m)
            nvol = m.numel() // (D * H * W)
            lib = load()
            ws = None
            if self.spline_order == 3:
                ws = ops.workspace(nvol * D * H * W * 8, m.device, "elastic")
            check(lib.mis_aug_map_coordinates(m.data_ptr(), out.data_ptr(), None if ws is None else ws.data_ptr(), nvol, D, H, W,
                                              None if fz is None else fz.data_ptr(), fy.data_ptr(), fx.data_ptr(), float(self.alpha),
                                              self.spline_order, m.element_size(), stream_ptr()), "mis_aug_map_coordinates")
            return out
        return m


class Standardize:
    """transforms.py:495-523: (m - mean) / clip(std, eps). Volume statistics are reduced on the device when not given; channelwise=True: per slice of the first
    axis (the reference's `axes[1:]`), i.e. per channel of a (C, D, H, W) volume."""

    def __init__(self, eps=1e-10, mean=None, std=None, channelwise=False, **kwargs):
        if mean is not None or std is not None:
            assert mean is not None and std is not None
        self.mean, self.std, self.eps, self.channelwise = mean, std, eps, channelwise

    def _one(self, m, mean, std):
        if mean is None:
            n = m.numel()
            if n % 4:
                raise MisError("Standardize: volume size must be a multiple of 4")
            s = torch.zeros(1, 4, device=m.device)
            q = torch.zeros(1, 4, device=m.device)
            ops.chanstats(m.reshape(1, 1, 1, n // 4, 4), s, q)
            mean = s.double().sum().item() / n
            # two passes like np.std (E[x^2] - mean^2 from fp32 partial sums cancels catastrophically when |mean| >> std): deviations first, then their moments
            dev_ = torch.empty_like(m)
            check(load().mis_aug_pointwise(m.data_ptr(), dev_.data_ptr(), n, 1.0, -mean, 0, 0.0, 0.0, 0.0, 0, stream_ptr()), "mis_aug_pointwise")
            ops.chanstats(dev_.reshape(1, 1, 1, n // 4, 4), s, q)
            d1, d2 = s.double().sum().item() / n, q.double().sum().item() / n
            mean += d1
            std = max(d2 - d1 * d1, 0.0) ** 0.5
        std = max(float(std), self.eps)
        out = torch.empty_like(m)
        a = 1.0 / std
        check(load().mis_aug_pointwise(m.data_ptr(), out.data_ptr(), m.numel(), a, -float(mean) * a, 0, 0.0, 0.0, 0.0, 0, stream_ptr()), "mis_aug_pointwise")
        return out

    def __call__(self, m):
        m = _dev(m)
        if m.dtype != torch.float32:
            raise MisError("Standardize: fp32 volumes only")
        if self.mean is not None:          # given statistics win over channelwise, as in the reference (:509-510); per-channel sequences broadcast over the first axis
            mean, std = np.asarray(self.mean, dtype=np.float64).reshape(-1), np.asarray(self.std, dtype=np.float64).reshape(-1)
            if mean.size == 1 and std.size == 1:
                return self._one(m.contiguous(), float(mean[0]), float(std[0]))
            if mean.size != m.shape[0] or std.size != m.shape[0]:
                raise MisError("Standardize: mean / std must be scalars or one value per channel")
            return torch.stack([self._one(m[c].contiguous(), float(mean[c]), float(std[c])) for c in range(m.shape[0])])
        if self.channelwise:
            return torch.stack([self._one(m[c].contiguous(), None, None) for c in range(m.shape[0])])
        return self._one(m.contiguous(), None, None)


class Normalize:
    """transforms.py:547-605: min-max scaling to [-1, 1] (or [0, 1] with norm01) with clipping.  Bounds that are not given come from the data (np.min / np.max,
    reduced on the device by mis_minmax); channelwise=True: bounds per slice of the first axis, given as lists in which the string 'None' (or None) means
    "from the data", as in the reference."""

    def __init__(self, min_value=None, max_value=None, norm01=False, channelwise=False, eps=1e-10, **kwargs):
        if min_value is not None and max_value is not None and not channelwise:
            assert max_value > min_value
        self.min_value, self.max_value, self.norm01, self.channelwise, self.eps = min_value, max_value, norm01, channelwise, eps

    @staticmethod
    def _minmax(m):
        ws = ops.workspace(2 * 1024 * 4, m.device, "minmax")
        out = torch.empty(2, device=m.device)
        check(load().mis_minmax(m.data_ptr(), m.numel(), ws.data_ptr(), out.data_ptr(), stream_ptr()), "mis_minmax")
        lo, hi = out.cpu().tolist()
        return lo, hi

    def _one(self, m, lo, hi):
        if lo is None or hi is None:
            dlo, dhi = self._minmax(m)
            lo = dlo if lo is None else lo
            hi = dhi if hi is None else hi
        out = torch.empty_like(m)
        a = 1.0 / (float(hi) - float(lo) + self.eps)
        lib = load()
        if abs(float(lo)) > 2.0 * (float(hi) - float(lo)):
            # bounds far from 0 relative to their distance (data-derived bounds of an offset channel): a * m - a * lo would cancel in fp32; subtract first, as numpy does
            check(lib.mis_aug_pointwise(m.data_ptr(), out.data_ptr(), m.numel(), 1.0, -float(lo), 0, 0.0, 0.0, 0.0, 0, stream_ptr()), "mis_aug_pointwise")
            m, b = out, 0.0
        else:
            b = -float(lo) * a
        args = (a, b, 1, 0.0, 1.0) if self.norm01 else (2 * a, 2 * b - 1, 1, -1.0, 1.0)
        check(lib.mis_aug_pointwise(m.data_ptr(), out.data_ptr(), m.numel(), *args, 0.0, 0, stream_ptr()), "mis_aug_pointwise")
        return out

    def __call__(self, m):
        m = _dev(m)
        if m.dtype != torch.float32:
            raise MisError("Normalize: fp32 volumes only")
        if not self.channelwise:
            return self._one(m.contiguous(), self.min_value, self.max_value)

        def per_channel(v):
            if v is None:
                return [None] * m.shape[0]
            v = list(v)
            if len(v) != m.shape[0]:
                raise MisError("Normalize(channelwise=True): one bound per channel expected")
            return [None if (x is None or x == "None") else float(x) for x in v]

        los, his = per_channel(self.min_value), per_channel(self.max_value)
        return torch.stack([self._one(m[c].contiguous(), los[c], his[c]) for c in range(m.shape[0])])


_POLY_DEV = {}
_PINNED = []


def _pinned_key(key):
    """a pinned staging copy of a generator key (ring of 8 buffers: a buffer is reused long after the stream-ordered upload that read it has run)"""
    if not _PINNED:
        _PINNED.extend([torch.empty(624, dtype=torch.int32).pin_memory() for _ in range(8)] + [0])
    i = _PINNED[-1]
    _PINNED[-1] = (i + 1) % 8
    buf = _PINNED[i]
    buf.copy_(torch.from_numpy(key.view(np.int32)))
    return buf


def _serial_legacy_normal(random_state, m, std):
    """the round-2 path: ONE workgroup generates the words, one draws the normals (~10 ms per 128^3); kept as the A/B arm and cross-check of the parallel path
    (AdditiveGaussianNoise(exact="serial"))"""
    lib = load()
    name, key, pos, has_gauss, cached = random_state.get_state()
    count = m.numel()
    flat = m.reshape(-1)
    out = torch.empty_like(flat)
    pairs = (count - int(has_gauss) + 1) // 2
    attempts = int(pairs * 1.2733 + 6.0 * math.sqrt(max(pairs, 1) * 0.274) + 64)     # acceptance pi/4 per attempt, six sigma of slack
    while True:
        key_d = torch.from_numpy(np.ascontiguousarray(key, dtype=np.uint32).view(np.int32)).to(m.device)
        pos_d = torch.tensor([int(pos)], dtype=torch.int32, device=m.device)
        words = torch.empty(4 * attempts, dtype=torch.int32, device=m.device)
        res = torch.zeros(5, dtype=torch.int64, device=m.device)
        check(lib.mis_mt19937_words(key_d.data_ptr(), pos_d.data_ptr(), words.data_ptr(), 4 * attempts, stream_ptr()), "mis_mt19937_words")
        check(lib.mis_legacy_normal(words.data_ptr(), attempts, flat.data_ptr(), out.data_ptr(), count, float(std), int(has_gauss), float(cached), res.data_ptr(),
                                    stream_ptr()), "mis_legacy_normal")
        r = res.cpu().numpy()
        if int(r[4]) == 1:
            break
        attempts *= 2                                     # (practically unreachable) the word stream ran out: start again from the same state with more
    used = int(r[0])
    if used:
        random_state.randint(0, 4294967296, size=4 * used, dtype=np.uint32)      # one 32-bit word each: the same 4 * used words the device consumed
    st = random_state.get_state()
    random_state.set_state((st[0], st[1], st[2], int(r[1]), float(np.int64(r[2]).view(np.float64)) if int(r[1]) else 0.0))
    return out.view(m.shape)


class _PendingState:
    """what the device still owes the host RandomState after a deferred exact-noise call: result5 (attempts used, gauss cache) and the key of the block in which the
    stream position ends, both still in HBM.  resolve() reads them (2.6 KB; by the next call the kernels have long finished, so nothing stalls) and sets the state."""

    def __init__(self, random_state, name, key, pos, res, raw):
        self.rs, self.name, self.key, self.pos, self.res, self.raw = random_state, name, key, pos, res, raw
        # the kernels that produce res / raw were enqueued on the CURRENT stream: resolve() may run under another one, so it waits for this event, not for
        # whatever stream happens to be current then
        self.event = torch.cuda.Event()
        self.event.record(torch.cuda.current_stream(res.device))

    def resolve(self):
        self.event.synchronize()
        r = self.res.cpu().numpy()
        if int(r[4]) != 1:
            raise MisError("exact Gaussian noise: the pre-drawn word stream was too short (eight sigma of slack) - the field of the previous call is incomplete; "
                           "use defer_state=False")
        used = int(r[0])
        key, pos = self.key, self.pos
        if used:
            end = pos + 4 * used
            blk, pos = (end // 624 - 1, 624) if end % 624 == 0 else (end // 624, end % 624)
            if blk > 0:
                key = self.raw.cpu().numpy().view(np.uint32)
        self.rs.set_state((self.name, key, pos, int(r[1]), float(np.int64(r[2]).view(np.float64)) if int(r[1]) else 0.0))


def _legacy_normal_on_device(random_state, m, std, defer=False):
    """m + random_state.normal(0, std, size=m.shape) with the field drawn ON THE DEVICE from numpy's own stream (csrc/mt19937.hip), by many workgroups: the word stream
    is cut into up to 64 chunks whose start keys come from GF(2) jump-ahead (mt_jump.py: one level of independent jumps from the current key), every chunk generates its words, and the polar
    rejection loop of numpy's legacy_gauss becomes count / scan / write passes.  The host RandomState ends exactly where the reference leaves it: key of the block in
    which the last consumed word lies (read back: 2.5 KB), position, gauss cache.  defer=True: that read-back is left to the caller (returns (field, _PendingState)):
    no host synchronisation inside the call."""
    from . import mt_jump
    lib = load()
    name, key, pos, has_gauss, cached = random_state.get_state()
    key = np.ascontiguousarray(key, dtype=np.uint32)
    pos = int(pos)
    dev = m.device
    count = m.numel()
    flat = m.reshape(-1)
    out = torch.empty_like(flat)
    pairs0 = (count + 1) // 2                                 # (the chunking must not depend on the gauss cache or on pos: its jump masks are cached per size)
    attempts = max(int(pairs0 * 1.2733 + 8.0 * math.sqrt(max(pairs0, 1) * 0.274) + 64), 1)     # acceptance pi/4 per attempt, eight sigma of slack
    while True:
        lo, hi = pos, pos + 4 * attempts                      # stream positions, counted from key[0]
        nblocks = (624 + 4 * attempts + 623) // 624           # blocks touched for the worst pos
        want = 1                                              # chunks: a power of two, at most 64, at least 8 blocks each
        while want < 64 and nblocks > 8 * want:
            want *= 2
        J = 624 * ((nblocks + want - 1) // want)
        nchunks = (nblocks * 624 + J - 1) // J
        states = torch.empty(nchunks, 624, dtype=torch.int32, device=dev)
        stage = _pinned_key(key)                              # pinned: the upload is stream-ordered and does not block the host
        states[0].copy_(stage, non_blocking=True)
        if nchunks > 1:
            pk = (J, nchunks, str(dev))
            if pk not in _POLY_DEV:                           # ~25 ms per chunk on the host, once per volume size
                _POLY_DEV[pk] = torch.from_numpy(mt_jump.jump_polys_from_start(J, nchunks).view(np.int32)).to(dev)
            # every chunk key straight from states[0]: ONE level of independent jumps, each shared by 8 workgroups
            check(lib.mis_mt_jump(states.data_ptr(), 0, 1, nchunks - 1, _POLY_DEV[pk].data_ptr(), 0, 1, 8, stream_ptr()), "mis_mt_jump")
        words = torch.empty(4 * attempts, dtype=torch.int32, device=dev)
        check(lib.mis_mt_generate(states.data_ptr(), nchunks, J, lo, hi, words.data_ptr(), -1, None, None, 0, stream_ptr()), "mis_mt_generate")
        res = torch.zeros(5, dtype=torch.int64, device=dev)
        ws = ops.workspace(lib.mis_legacy_normal_par_workspace_bytes(attempts), dev, "legacy_normal")
        check(lib.mis_legacy_normal_par(words.data_ptr(), attempts, flat.data_ptr(), out.data_ptr(), count, float(std), int(has_gauss), float(cached), ws.data_ptr(),
                                        res.data_ptr(), stream_ptr()), "mis_legacy_normal_par")
        # the key of the block in which the position ends, chosen ON THE DEVICE from the attempts used
        raw = torch.empty(624, dtype=torch.int32, device=dev)
        check(lib.mis_mt_generate(states.data_ptr(), nchunks, J, 0, 0, None, -1, raw.data_ptr(), res.data_ptr(), pos, stream_ptr()), "mis_mt_generate")
        pending = _PendingState(random_state, name, key, pos, res, raw)
        pending.keep = (stage, states, words)                 # alive until the kernels that read them have run
        if defer:
            return out.view(m.shape), pending
        if int(res[4].item()) == 1:
            break
        attempts *= 2                                         # (practically unreachable) the word stream ran out: start again from the same state with more
    pending.resolve()
    return out.view(m.shape)


class AdditiveGaussianNoise:
    """transforms.py:608-619: if uniform() < p: std = uniform(scale); m + N(0, std).
    exact=True / "device" (the DEFAULT since round 3: a drop-in must give the reference's results): the reference's OWN field - numpy's MT19937 + legacy polar
    Box-Muller reproduced on the device, bit-comparable with the reference (golden g16_gauss_noise.npz), the RandomState left where the reference leaves it;
    exact=False: opt-in fast path, the counter-based generator of augment.hip (same distribution, one fused pass, a different field);
    exact="host": numpy on the host + upload (the round-1 parity mode); exact="serial": the round-2 single-workgroup device path (A/B arm)."""

    def __init__(self, random_state, scale=(0.0, 1.0), execution_probability=0.1, exact=True, defer_state=False, **kwargs):
        """defer_state (exact device path only): the read-back that advances `random_state` to where the reference leaves it is postponed to the next call of this
        transform (or flush()), so that the call itself never synchronises with the device.  Only for a RandomState this transform owns - `Transformer` builds one per
        transform (transforms.py:751) and switches it on; with a RandomState shared between transforms keep the default."""
        self.execution_probability = execution_probability
        self.random_state = random_state
        self.scale = scale
        self.exact = exact
        self.defer_state = defer_state
        self._pending = None

    def flush(self):
        """bring `random_state` up to date (deferred mode)"""
        if self._pending is not None:
            p, self._pending = self._pending, None
            p.resolve()

    def __call__(self, m):
        self.flush()
        if self.random_state.uniform() < self.execution_probability:
            std = self.random_state.uniform(self.scale[0], self.scale[1])
            m = _dev(m)
            if self.exact == "host":
                noise = self.random_state.normal(0, std, size=tuple(m.shape))
                return (m.double() + torch.from_numpy(noise).to(m.device)).float()   # upload of a host-generated field
            if self.exact == "serial":
                return _serial_legacy_normal(self.random_state, m.contiguous().float(), std)
            if self.exact:
                if self.defer_state:
                    out, self._pending = _legacy_normal_on_device(self.random_state, m.contiguous().float(), std, defer=True)
                    return out
                return _legacy_normal_on_device(self.random_state, m.contiguous().float(), std)
            seed = int(self.random_state.randint(0, 2 ** 31 - 1))
            out = torch.empty_like(m)
            check(load().mis_aug_pointwise(m.data_ptr(), out.data_ptr(), m.numel(), 1.0, 0.0, 0, 0.0, 0.0, float(std), seed, stream_ptr()),
                  "mis_aug_pointwise")
            return out
        return m


class ToTensor:
    """transforms.py:636-655: add the channel axis for 3-D inputs and cast; the data already lives on the device."""

    def __init__(self, expand_dims, dtype=np.float32, **kwargs):
        self.expand_dims = expand_dims
        self.dtype = dtype

    def __call__(self, m):
        m = _dev(m)
        if self.expand_dims and m.dim() == 3:
            m = m.unsqueeze(0)
        td = {np.float32: torch.float32, np.int64: torch.int64, "float32": torch.float32, "int64": torch.int64}.get(self.dtype, torch.float32)
        return m.to(td)


class LabelToTensor:
    def __call__(self, m):
        return _dev(m).to(torch.int64)


def _unbuilt(name):
    class _U:
        def __init__(self, *a, **k):
            raise NotImplementedError(f"{name}: not built on the device (SURVEY.md §8f2)")
    _U.__name__ = name
    return _U


class AdditivePoissonNoise:
    """transforms.py:622-633: if uniform() < p: lam = uniform(lam range); m + Poisson(lam) field.  The field is drawn from the reference's
    own stream on the host (keeps every later draw aligned) and added on the device; fp32 result = the reference's float64 sum rounded once."""

    def __init__(self, random_state, lam=(0.0, 1.0), execution_probability=0.1, **kwargs):
        self.execution_probability = execution_probability
        self.random_state = random_state
        self.lam = lam

    def __call__(self, m):
        if self.random_state.uniform() < self.execution_probability:
            lam = self.random_state.uniform(self.lam[0], self.lam[1])
            m = _dev(m).to(torch.float32)
            noise = torch.from_numpy(self.random_state.poisson(lam, size=tuple(m.shape)).astype(np.float32)).to(m.device)
            if m.numel() % 4 == 0:
                out = torch.empty_like(m)
                ops.add_act(m.view(1, 1, -1, 4), noise.view(1, 1, -1, 4), out.view(1, 1, -1, 4))
                return out
            return m + noise
        return m


class CropToFixed:
    """transforms.py:194-247: random (or centred) (crop_y, crop_x) window of the last two axes; an axis shorter than the crop is taken whole
    and mirror-padded (numpy 'reflect') to the crop size.  One gather kernel (mis_aug_crop_reflect); raw fp32 and int64 labels alike."""

    def __init__(self, random_state, size=(256, 256), centered=False, **kwargs):
        self.random_state = random_state
        self.crop_y, self.crop_x = size
        self.centered = centered

    @staticmethod
    def _start_pad(crop_size, max_size, rs):
        if crop_size < max_size:
            return (rs.randint(max_size - crop_size) if rs is not None else (max_size - crop_size) // 2), 0
        if rs is not None:
            rs.randint(1)                                   # the reference still draws (range 1) - keeps the stream aligned
        return 0, (crop_size - max_size) // 2

    def __call__(self, m):
        m = _dev(m)
        y, x = m.shape[-2:]
        rs = None if self.centered else self.random_state
        y_start, y_lo = self._start_pad(self.crop_y, y, rs)
        x_start, x_lo = self._start_pad(self.crop_x, x, rs)
        es = m.element_size()
        if es not in (4, 8):
            raise MisError(f"augment: unsupported element size {es} ({m.dtype})")
        out = torch.empty(tuple(m.shape[:-2]) + (self.crop_y, self.crop_x), dtype=m.dtype, device=m.device)
        check(load().mis_aug_crop_reflect(m.data_ptr(), out.data_ptr(), m.numel() // (y * x), y, x, y_start - y_lo, x_start - x_lo,
                                          self.crop_y, self.crop_x, es, stream_ptr()), "mis_aug_crop_reflect")
        return out


class PercentileNormalizer:
    """transforms.py:526-543: (m - P_pmin) / (P_pmax - P_pmin + eps) with numpy's linear-interpolated percentiles.  The two order statistics each
    percentile needs are SELECTED on the device (torch.kthvalue on the flattened volume - the one place where a library selection primitive is used),
    the interpolation follows numpy's `_lerp`, the normalisation is one elementwise kernel."""

    def __init__(self, pmin=1, pmax=99.6, channelwise=False, eps=1e-10, **kwargs):
        self.eps, self.pmin, self.pmax, self.channelwise = eps, pmin, pmax, channelwise

    @staticmethod
    def _percentile(flat, q):
        n = flat.numel()
        pos = (n - 1) * (q / 100.0)
        lo = int(np.floor(pos))
        hi = min(lo + 1, n - 1)
        t = np.float32(pos - lo)
        a = flat.kthvalue(lo + 1).values.item()
        b = flat.kthvalue(hi + 1).values.item()
        a, b = np.float32(a), np.float32(b)
        d = b - a
        return np.float32(b - d * (np.float32(1) - t)) if t >= 0.5 else np.float32(a + d * t)      # numpy _lerp

    def __call__(self, m):
        m = _dev(m)
        if m.dtype != torch.float32:
            raise MisError("PercentileNormalizer: fp32 volumes only")
        if self.channelwise:          # transforms.py:534-539: percentiles over every axis but the first, i.e. each channel normalised by its own pair (round 4)
            return torch.stack([self._normalize(m[c].contiguous()) for c in range(m.shape[0])])
        return self._normalize(m)

    def _normalize(self, m):
        flat = m.reshape(-1)
        pmin, pmax = self._percentile(flat, self.pmin), self._percentile(flat, self.pmax)
        a = np.float32(1.0) / np.float32(pmax - pmin + np.float32(self.eps))
        out = torch.empty_like(m)
        check(load().mis_aug_pointwise(m.data_ptr(), out.data_ptr(), m.numel(), float(a), float(-pmin * a), 0, 0.0, 0.0, 0.0, 0, stream_ptr()),
              "mis_aug_pointwise")
        return out


class Identity:
    """transforms.py:686-691"""

    def __init__(self, **kwargs):
        pass

    def __call__(self, m):
        return m


class GaussianBlur3D:
    """transforms.py:708-718: with probability p (Python's global `random`, like the reference), skimage.filters.gaussian(x, sigma ~ U(sigma[0], sigma[1])).
    For a float volume that is scipy.ndimage.gaussian_filter(x, sigma, mode='nearest', truncate=4.0) in the volume's dtype: three separable passes on the
    device (csrc/augment.hip aug_gauss1d_kernel<float, NEAREST>), each rounded to fp32 like scipy's per-axis output array.  A 4-D input (C, D, H, W) is
    what skimage would treat as one 4-D image (it would also blur ACROSS channels); the reference only applies it to 3-D raw patches and so does this."""

    def __init__(self, sigma=[.1, 2.], execution_probability=0.5, **kwargs):
        self.sigma = sigma
        self.execution_probability = execution_probability

    def __call__(self, x):
        import random
        if random.random() < self.execution_probability:
            sigma = random.uniform(self.sigma[0], self.sigma[1])
            return self.blur(x, sigma)
        return x

    @staticmethod
    def blur(x, sigma):
        x = _dev(x)
        if x.dim() != 3:
            raise MisError(f"GaussianBlur3D: a (D, H, W) volume is expected, got {tuple(x.shape)}")
        a = x.to(torch.float32).contiguous()
        sd = float(sigma)
        radius = int(4.0 * sd + 0.5)
        xs = np.arange(-radius, radius + 1)
        phi = np.exp(-0.5 / (sd * sd) * xs ** 2)
        phi = (phi / phi.sum())[::-1].copy()
        wdev = torch.from_numpy(phi).to(a.device)
        D, H, W = a.shape
        b = torch.empty_like(a)
        src = a
        for ax in range(3):
            dst = b if src is not b else torch.empty_like(a)
            check(load().mis_aug_gauss1d_f32(src.data_ptr(), dst.data_ptr(), 1, D, H, W, ax, wdev.data_ptr(), radius, 1, stream_ptr()), "mis_aug_gauss1d_f32")
            src = dst
        return src


# label -> boundary / affinity targets and connected-component relabelling need skimage (find_boundaries, measure.label): names kept, construction raises
for _n in ("AbstractLabelToBoundary", "StandardLabelToBoundary", "BlobsToMask", "RandomLabelToAffinities", "LabelToAffinities", "LabelToZAffinities",
           "LabelToBoundaryAndAffinities", "LabelToMaskAndAffinities", "Relabel", "RgbToLabel"):
    globals()[_n] = _unbuilt(_n)


class Transformer:
    """transforms.py:721-753."""

    def __init__(self, phase_config, base_config):
        self.phase_config = phase_config
        self.config_base = base_config
        self.seed = GLOBAL_RANDOM_STATE.randint(10000000)

    def raw_transform(self):
        return self._create_transform('raw')

    def label_transform(self):
        return self._create_transform('label')

    def weight_transform(self):
        return self._create_transform('weight')

    @staticmethod
    def _transformer_class(class_name):
        m = importlib.import_module(__name__)
        return getattr(m, class_name)

    def _create_transform(self, name):
        assert name in self.phase_config, f'Could not find {name} transform'
        return Compose([self._create_augmentation(c) for c in self.phase_config[name]])

    def _create_augmentation(self, c):
        config = dict(self.config_base)
        config.update(c)
        config['random_state'] = np.random.RandomState(self.seed)
        aug_class = self._transformer_class(config['name'])
        if aug_class is AdditiveGaussianNoise:
            config.setdefault('defer_state', True)          # the RandomState above belongs to this transform alone: its read-back can wait for the next call
        return aug_class(**config)
