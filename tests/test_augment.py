"""3-D augmentation: the numpy oracle against golden vectors from the real reference classes (CPU), and the on-device
transforms (HIP gather / elementwise kernels behind the reference's class names) against both (GPU)."""
import numpy as np
import pytest
import torch

from conftest import load_golden
from oracle import augment_oracle as ao

CROPS = [((8, 9), False, "v"), ((13, 20), False, "v"), ((7, 19), False, "v"), ((6, 6), True, "v"), ((12, 31), True, "v"), ((4, 9), False, "c4"),
         ((8, 8), False, "label")]


def _replay_single(g):
    """replays the explicit-seed single-transform goldens with the oracle"""
    v, sq = g["v"], g["sq"]
    for s in range(8):
        rs = np.random.RandomState(100 + s)
        axes = [(1, 0), (2, 1), (2, 0)]
        axis = axes[rs.randint(len(axes))]
        angle = rs.randint(-30, 30)
        yield f"rot_{s}", ao.rotate0(v, angle, axis)
    for s in range(4):
        rs = np.random.RandomState(200 + s)
        mask = 0
        for axis in (0, 1, 2):
            if rs.uniform() > 0.5:
                mask |= 1 << axis
        yield f"flip_{s}", ao.flip(v, mask)
    for s in range(6):
        yield f"rot90_{s}", ao.rot90(sq, np.random.RandomState(300 + s).randint(0, 4))


def test_oracle_matches_reference_goldens():
    g = load_golden("g5_augment.npz")
    for name, got in _replay_single(g):
        assert np.array_equal(got, g[name]), name
    assert int(g["transformer_seed"]) == 889991
    p = ao.Pipeline(int(g["transformer_seed"]), [[2, 1]], 0.6, mean=0.05)
    for i in range(6):
        assert np.array_equal(p.raw(g["raw"]), g[f"pipe_raw_{i}"]), f"raw {i}"
        assert np.array_equal(p.label(g["label"]), g[f"pipe_label_{i}"]), f"label {i}"
    assert np.allclose(ao.standardize(g["v"]), g["std_auto"], atol=1e-6)
    for s in range(6):      # cubic-spline rotation: float64 spline coefficients, result equal up to float32 rounding
        rs = np.random.RandomState(400 + s)
        axis = [(1, 0), (2, 1), (2, 0)][rs.randint(3)]
        assert np.allclose(ao.rotate3(g["v"], rs.randint(-30, 30), axis), g[f"rot3_{s}"], rtol=0, atol=5e-7), f"rot3_{s}"
    rs = np.random.RandomState(9)
    rs.uniform()
    assert np.array_equal(ao.contrast(g["v"], 0.05, rs.uniform(0.5, 1.5)), g["contrast"])
    for key, src, order, seed, kw in (("el_raw_out", "el_raw", 3, 500, dict(alpha=15, sigma=3)), ("el_label_out", "el_label", 0, 500, dict(alpha=15, sigma=3)),
                                      ("el_raw_2d", "el_raw", 3, 501, dict(alpha=2000, sigma=50, apply_3d=False))):
        rs = np.random.RandomState(seed)
        rs.uniform()
        assert np.array_equal(ao.elastic(g[src], rs, order, **kw), g[key]), key
    for s_, (size, cen, src) in enumerate(CROPS):
        assert np.array_equal(ao.crop_to_fixed(g[src], np.random.RandomState(600 + s_), size, cen), g[f"crop_{s_}"]), f"crop_{s_}"
    rs = np.random.RandomState(700)
    rs.uniform()
    assert np.array_equal(ao.poisson_noise(g["v"], rs, (0.5, 3.0)), g["poisson"])


@pytest.mark.gpu
def test_device_crop_and_poisson_match_reference_goldens():
    from mdeical_image_segmentation_amd.augment.unet3d_augment import transforms as tr
    g = load_golden("g5_augment.npz")
    for s_, (size, cen, src) in enumerate(CROPS):
        out = tr.CropToFixed(np.random.RandomState(600 + s_), size=size, centered=cen)(g[src])
        assert out.dtype == torch.from_numpy(g[src]).dtype and np.array_equal(out.cpu().numpy(), g[f"crop_{s_}"]), f"crop_{s_}"
    out = tr.AdditivePoissonNoise(np.random.RandomState(700), lam=(0.5, 3.0), execution_probability=1.0)(g["v"])
    assert np.array_equal(out.cpu().numpy(), g["poisson"].astype(np.float32))
    for key, kw in (("pnorm", {}), ("pnorm_5_90", dict(pmin=5, pmax=90))):
        got = tr.PercentileNormalizer(**kw)(g["v"]).cpu().numpy()
        assert np.abs(got - g[key]).max() < 2e-6 * max(1.0, np.abs(g[key]).max()), key
    # channelwise=True (transforms.py:534-539, round 4): every channel by its own percentile pair - against the reference's own three numpy lines on a 4-D input
    v4 = np.stack([g["v"], g["v"][::-1].copy() * 2.5 - 0.3, np.sqrt(np.abs(g["v"]))]).astype(np.float32)
    axes = tuple(range(1, v4.ndim))
    for kw in ({}, dict(pmin=5, pmax=90)):
        lo = np.percentile(v4, kw.get("pmin", 1), axis=axes, keepdims=True)
        hi = np.percentile(v4, kw.get("pmax", 99.6), axis=axes, keepdims=True)
        want = (v4 - lo) / (hi - lo + 1e-10)
        got = tr.PercentileNormalizer(channelwise=True, **kw)(v4).cpu().numpy()
        assert got.shape == want.shape and np.abs(got - want).max() < 2e-6 * max(1.0, np.abs(want).max()), kw
    keep = tr.AdditivePoissonNoise(np.random.RandomState(700), execution_probability=0.0)(g["v"])
    assert np.array_equal(np.asarray(keep if isinstance(keep, np.ndarray) else keep.cpu().numpy()), g["v"])


@pytest.mark.gpu
def test_device_transforms_match_reference_goldens():
    from mdeical_image_segmentation_amd.augment.unet3d_augment import transforms as tr
    g = load_golden("g5_augment.npz")
    v, sq = g["v"], g["sq"]
    for s in range(8):
        out = tr.RandomRotate(np.random.RandomState(100 + s), angle_spectrum=30, mode="reflect", order=0)(v)
        assert np.array_equal(out.cpu().numpy(), g[f"rot_{s}"]), f"rot_{s}"
    for s in range(4):
        assert np.array_equal(tr.RandomFlip(np.random.RandomState(200 + s))(v).cpu().numpy(), g[f"flip_{s}"]), f"flip_{s}"
    for s in range(6):
        assert np.array_equal(tr.RandomRotate90(np.random.RandomState(300 + s))(sq).cpu().numpy(), g[f"rot90_{s}"]), f"rot90_{s}"
    assert np.array_equal(tr.RandomFlip(np.random.RandomState(7))(g["c4"]).cpu().numpy(), g["c4_flip"])
    assert np.array_equal(tr.RandomRotate(np.random.RandomState(8), axes=[(2, 1)])(g["c4"]).cpu().numpy(), g["c4_rot"])
    for s in range(6):
        out = tr.RandomRotate(np.random.RandomState(400 + s), angle_spectrum=30, mode="reflect", order=3)(v)
        assert np.allclose(out.cpu().numpy(), g[f"rot3_{s}"], rtol=0, atol=5e-7), f"rot3_{s}"
    assert np.allclose(tr.RandomRotate(np.random.RandomState(8), axes=[(2, 1)], order=3)(g["c4"]).cpu().numpy(), g["c4_rot3"], rtol=0, atol=5e-7)
    # elastic deformation: labels (order 0) exact, raw (order 3) up to fp32 rounding of float64 spline arithmetic
    el = tr.ElasticDeformation(np.random.RandomState(500), spline_order=0, alpha=15, sigma=3, execution_probability=1.0)(g["el_label"])
    assert np.array_equal(el.cpu().numpy(), g["el_label_out"])
    er = tr.ElasticDeformation(np.random.RandomState(500), spline_order=3, alpha=15, sigma=3, execution_probability=1.0)(g["el_raw"])
    assert np.abs(er.cpu().numpy() - g["el_raw_out"]).max() < 1e-6
    e2 = tr.ElasticDeformation(np.random.RandomState(501), spline_order=3, alpha=2000, sigma=50, execution_probability=1.0, apply_3d=False)(g["el_raw"])
    assert np.abs(e2.cpu().numpy() - g["el_raw_2d"]).max() < 1e-6
    keep = tr.ElasticDeformation(np.random.RandomState(1), spline_order=3, execution_probability=0.0)(g["el_raw"])
    assert np.array_equal(np.asarray(keep if isinstance(keep, np.ndarray) else keep.cpu().numpy()), g["el_raw"])
    assert np.array_equal(tr.RandomContrast(np.random.RandomState(9), mean=0.05, execution_probability=1.0)(v).cpu().numpy(), g["contrast"])
    assert np.allclose(tr.Standardize()(v).cpu().numpy(), g["std_auto"], atol=2e-6)
    assert np.allclose(tr.Standardize(mean=0.1, std=0.5)(v).cpu().numpy(), g["std_fixed"], atol=1e-6)
    assert np.allclose(tr.Normalize(min_value=-1.0, max_value=1.0)(v).cpu().numpy(), g["norm"], atol=1e-6)
    # the Transformer, seeded like the reference's first Transformer of a process, raw and label in lock-step
    tr.GLOBAL_RANDOM_STATE = np.random.RandomState(47)
    axes = [[2, 1]]
    cfg = {"raw": [{"name": "RandomFlip"}, {"name": "RandomRotate90"},
                   {"name": "RandomRotate", "axes": axes, "angle_spectrum": 30, "mode": "reflect", "order": 0},
                   {"name": "RandomContrast", "execution_probability": 0.6}],
           "label": [{"name": "RandomFlip"}, {"name": "RandomRotate90"},
                     {"name": "RandomRotate", "axes": axes, "angle_spectrum": 30, "mode": "reflect", "order": 0}]}
    t = tr.Transformer(cfg, {"mean": 0.05, "std": 1.0})
    assert t.seed == int(g["transformer_seed"])
    rt, lt = t.raw_transform(), t.label_transform()
    raw_d = torch.from_numpy(g["raw"]).cuda()
    lab_d = torch.from_numpy(g["label"]).cuda()
    for i in range(6):
        assert np.array_equal(rt(raw_d).cpu().numpy(), g[f"pipe_raw_{i}"]), f"pipe raw {i}"
        assert np.array_equal(lt(lab_d).cpu().numpy(), g[f"pipe_label_{i}"]), f"pipe label {i}"


@pytest.mark.gpu
def test_device_rotate_at_full_size_matches_oracle_and_noise_statistics():
    from mdeical_image_segmentation_amd.augment.unet3d_augment import transforms as tr
    rng = np.random.RandomState(1)
    vol = rng.rand(32, 128, 128).astype(np.float32)
    for seed in (1, 2, 3):
        rs = np.random.RandomState(seed)
        axes = [(1, 0), (2, 1), (2, 0)]
        axis = axes[rs.randint(len(axes))]
        angle = rs.randint(-30, 30)
        out = tr.RandomRotate(np.random.RandomState(seed))(vol)
        assert np.array_equal(out.cpu().numpy(), ao.rotate0(vol, angle, axis)), (seed, axis, angle)
    z = torch.zeros(64, 64, 64, device="cuda")
    n = tr.AdditiveGaussianNoise(np.random.RandomState(3), scale=(0.5, 0.5), execution_probability=1.0)(z)
    assert abs(n.mean().item()) < 5e-3 and abs(n.std().item() - 0.5) < 5e-3
    for seed in (4, 5):                                   # order 3 at full plane size against the oracle (= scipy up to fp32 rounding)
        rs = np.random.RandomState(seed)
        axis = [(1, 0), (2, 1), (2, 0)][rs.randint(3)]
        angle = rs.randint(-30, 30)
        out = tr.RandomRotate(np.random.RandomState(seed), order=3)(vol).cpu().numpy()
        ref = ao.rotate3(vol, angle, axis)
        assert np.abs(out - ref).max() < 5e-7, (seed, axis, angle, np.abs(out - ref).max())
    with pytest.raises(NotImplementedError):
        tr.RandomRotate(np.random.RandomState(1), order=3, mode="constant")(vol)


ROTATE_MODES = ["reflect", "grid-mirror", "constant", "grid-constant", "nearest", "mirror", "wrap", "grid-wrap"]


def test_rotate_boundary_modes_oracle_equals_scipy():
    """oracle.augment_oracle.rotate0_modes (numpy restatement of scipy's map_coordinate + rounding + index extension) is bit-identical to the call the reference makes,
    scipy.ndimage.rotate(..., order=0, mode=mode, cval=-1) (transforms.py:109-111), for every boundary mode; sizes incl. an axis of length 1"""
    from scipy import ndimage
    from oracle import augment_oracle as ao
    rng = np.random.RandomState(0)
    for shape in [(5, 9, 12), (1, 7, 3)]:
        m = rng.randn(*shape).astype(np.float32)
        for mode in ROTATE_MODES:
            for axes in [(1, 0), (2, 1), (2, 0)]:
                for angle in [-30, -7, 0, 17, 29, 90]:
                    ref = ndimage.rotate(m, angle, axes=axes, reshape=False, order=0, mode=mode, cval=-1)
                    assert np.array_equal(ao.rotate0_modes(m, angle, axes, mode, -1), ref), (shape, mode, axes, angle)


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ROTATE_MODES)
def test_random_rotate_boundary_modes(mode):
    """RandomRotate(order=0, mode=...) on the device (mis_aug_rotate0_mode) == scipy.ndimage.rotate with the same draws: float32 raw volumes and int64 label volumes,
    every boundary mode scipy offers (round 4: only 'reflect' was built)"""
    import torch
    from scipy import ndimage
    from mdeical_image_segmentation_amd.augment.unet3d_augment import transforms as tr
    vol = (np.random.RandomState(22).rand(2, 12, 33, 20).astype(np.float32) * 3 - 1)
    lab = np.random.RandomState(23).randint(0, 5, (12, 33, 20)).astype(np.int64)
    for seed in range(6):
        rs = np.random.RandomState(seed)
        axes = [(1, 0), (2, 1), (2, 0)]
        axis = axes[rs.randint(len(axes))]
        angle = rs.randint(-30, 30)
        for m in (vol, lab):
            if m.ndim == 4:
                ref = np.stack([ndimage.rotate(c, angle, axes=axis, reshape=False, order=0, mode=mode, cval=-1) for c in m])
            else:
                ref = ndimage.rotate(m, angle, axes=axis, reshape=False, order=0, mode=mode, cval=-1)
            out = tr.RandomRotate(np.random.RandomState(seed), mode=mode, order=0)(torch.from_numpy(m).cuda()).cpu().numpy()
            assert np.array_equal(out, ref), (mode, seed, axis, angle, int((out != ref).sum()))


@pytest.mark.gpu
@pytest.mark.parametrize("order", [1, 2, 4, 5])
def test_random_rotate_other_spline_orders(order):
    """RandomRotate(order = 1, 2, 4, 5; mode 'reflect') against the call the reference makes (transforms.py:109-111: scipy.ndimage.rotate(m, angle, axes, reshape=False,
    order, mode, cval=-1)): the same random draws, float64 spline arithmetic rounded to fp32 once - equal up to that rounding; a line of length 1 along a rotation axis is
    its own coefficient"""
    from scipy import ndimage
    from mdeical_image_segmentation_amd.augment.unet3d_augment import transforms as tr
    vol = (np.random.RandomState(21).rand(20, 33, 47).astype(np.float32) * 3 - 1)
    for seed in (1, 2, 3, 4):
        rs = np.random.RandomState(seed)
        axes = [(1, 0), (2, 1), (2, 0)]
        axis = axes[rs.randint(len(axes))]
        angle = rs.randint(-30, 30)
        ref = ndimage.rotate(vol, angle, axes=axis, reshape=False, order=order, mode="reflect", cval=-1)
        out = tr.RandomRotate(np.random.RandomState(seed), order=order)(vol).cpu().numpy()
        err = np.abs(out - ref).max()
        assert err < 1e-6, (order, seed, axis, angle, err)
        assert (out == ref).mean() > 0.98, (order, seed, (out == ref).mean())
    c4 = np.random.RandomState(5).rand(2, 6, 9, 1).astype(np.float32)           # channel-first 4-D input, a rotation axis of length 1
    ref = np.stack([ndimage.rotate(c, 17, axes=(2, 1), reshape=False, order=order, mode="reflect", cval=-1) for c in c4])
    rs = np.random.RandomState(0)
    for seed in range(200):                                                   # a seed whose draw is (axis index 0 of the one given, angle 17)
        rs = np.random.RandomState(seed)
        if rs.randint(1) == 0 and rs.randint(-30, 30) == 17:
            out = tr.RandomRotate(np.random.RandomState(seed), axes=[(2, 1)], order=order)(c4).cpu().numpy()
            assert np.abs(out - ref).max() < 1e-6
            break
    else:
        raise AssertionError("no seed with angle 17 among 200")


@pytest.mark.gpu
def test_gaussian_noise_field_on_the_reference_stream():
    """a20: AdditiveGaussianNoise(exact=True) draws the reference's OWN noise field on the device (MT19937 + numpy's legacy polar Box-Muller, csrc/mt19937.hip):
    equal to the REAL reference class's output (g16_gauss_noise.npz, float64 rounded once to float32) for three consecutive calls on one RandomState - odd element
    counts carry the cached second value across calls - and the RandomState ends where the reference's does."""
    import torch

    from mdeical_image_segmentation_amd.augment.unet3d_augment import transforms as tr
    g = load_golden("g16_gauss_noise.npz")
    rs = np.random.RandomState(777)
    t = tr.AdditiveGaussianNoise(rs, scale=(0.05, 0.3), execution_probability=1.0, exact=True)
    for i in range(3):
        out = t(torch.from_numpy(g[f"in_{i}"]).cuda()).cpu().numpy()
        want = g[f"out_{i}"].astype(np.float32)
        neq = int((out != want).sum())
        # the device's double log / sqrt may differ from glibc's in the last bit; after the rounding to float32 that can only show in a vanishing fraction of samples
        assert neq <= max(1, out.size // 100000), (i, neq)
        assert np.abs(out.astype(np.float64) - g[f"out_{i}"]).max() < 1e-6
    assert np.array_equal(np.array([rs.uniform(), rs.uniform()]), g["next_uniform"]), "the RandomState is not where the reference leaves it"
    rs2 = np.random.RandomState(778)
    t2 = tr.AdditiveGaussianNoise(rs2, scale=(0.0, 1.0), execution_probability=0.0, exact=True)
    x = torch.from_numpy(g["in_0"]).cuda()
    assert torch.equal(t2(x), x) and rs2.uniform() == g["skip_next"][0]
    # a larger field (many 624-word blocks, several scan chunks) against the CPU restatement of the same stream
    from oracle import augment_oracle as ao
    rs3, rs4 = np.random.RandomState(5), np.random.RandomState(5)
    v = np.random.RandomState(1).rand(24, 40, 40).astype(np.float32)
    a = tr.AdditiveGaussianNoise(rs3, scale=(0.1, 0.2), execution_probability=1.0, exact=True)(torch.from_numpy(v).cuda()).cpu().numpy()
    b = ao.additive_gaussian_noise(v, rs4, (0.1, 0.2), 1.0).astype(np.float32)
    assert int((a != b).sum()) <= 1 and rs3.uniform() == rs4.uniform()


@pytest.mark.gpu
@pytest.mark.parametrize("sigma", [0.1, 0.45, 1.3, 2.0])
def test_gaussian_blur3d_matches_scipy(sigma):
    """GaussianBlur3D (reference transforms.py:708-718) against the oracle's scipy call: the kernel walks a symmetric kernel the way scipy's correlate1d does
    (centre tap, then pairs from the far end inwards, double accumulation, fp32 rounding after every axis), so the result is bit-equal"""
    import random

    from mdeical_image_segmentation_amd.augment.unet3d_augment.transforms import GaussianBlur3D
    v = np.random.RandomState(5).randn(12, 20, 17).astype(np.float32)
    want = ao.gaussian_blur3d(v, sigma)
    got = GaussianBlur3D.blur(torch.from_numpy(v).cuda(), sigma).cpu().numpy()
    assert got.dtype == np.float32 and np.array_equal(got, want), np.abs(got - want).max()
    # the two draws come from Python's global generator, like the reference's
    random.seed(11)
    t = GaussianBlur3D(sigma=[0.5, 1.5], execution_probability=1.0)
    out = t(torch.from_numpy(v).cuda()).cpu().numpy()
    random.seed(11)
    random.random()
    assert np.array_equal(out, ao.gaussian_blur3d(v, random.uniform(0.5, 1.5)))
    assert GaussianBlur3D(execution_probability=0.0)(v) is v


@pytest.mark.gpu
def test_standardize_normalize_options_vs_reference_golden():
    """Standardize(channelwise=True) and Normalize with data-derived / per-channel bounds (transforms.py:495-523, 547-605), goldens from the real classes
    (tests/golden/make_golden_orders.py)"""
    from conftest import load_golden
    from mdeical_image_segmentation_amd.augment.unet3d_augment import transforms as tr
    g = load_golden("g17_orders.npz")
    c4, v = torch.from_numpy(g["aug/c4"]).cuda(), torch.from_numpy(g["aug/v"]).cuda()
    # (numpy reduces in float32: one ulp of the mean of channel 2 - 6e-7 at |mean| = 5 - is 1e-5 standard deviations at std = 0.058; the device sums in float64)
    assert np.allclose(tr.Standardize(channelwise=True)(c4).cpu().numpy(), g["aug/std_channelwise"], atol=2e-5)
    assert np.allclose(tr.Standardize(channelwise=True)(c4)[:2].cpu().numpy(), g["aug/std_channelwise"][:2], atol=3e-6)
    assert np.allclose(tr.Normalize()(v).cpu().numpy(), g["aug/norm_data"], atol=1e-6)
    assert np.allclose(tr.Normalize(norm01=True)(v).cpu().numpy(), g["aug/norm_data01"], atol=1e-6)
    assert np.allclose(tr.Normalize(min_value=-1.0)(v).cpu().numpy(), g["aug/norm_min_only"], atol=1e-6)
    assert np.allclose(tr.Normalize(channelwise=True)(c4).cpu().numpy(), g["aug/norm_channelwise"], atol=1e-6)
    assert np.allclose(tr.Normalize(min_value=["None", -2.5, 5.0], channelwise=True)(c4).cpu().numpy(), g["aug/norm_channelwise_mixed"], atol=1e-6)


def test_mt19937_jump_ahead_polynomials_match_numpy():
    """host side of the parallel exact-noise path (augment/unet3d_augment/mt_jump.py): the characteristic polynomial from Berlekamp-Massey has degree 19937, and the
    key J words ahead computed as the GF(2) convolution with t^(J-1) mod phi equals the key numpy reaches by drawing J words - for J = 624 * 40 and its doublings, from
    a state in the middle of a block"""
    from mdeical_image_segmentation_amd.augment.unet3d_augment import mt_jump
    assert mt_jump.phi().bit_length() - 1 == 19937
    J = 624 * 40
    g = mt_jump.jump_polys(J, 3)
    rs = np.random.RandomState(2024)
    rs.randint(0, 2 ** 32, size=1000, dtype=np.uint32)
    st = rs.get_state()
    for k in range(3):
        rs2 = np.random.RandomState()
        rs2.set_state(st)
        rs2.randint(0, 2 ** 32, size=J << k, dtype=np.uint32)
        assert np.array_equal(rs2.get_state()[1], mt_jump.jump_key(st[1], g[k])), k
        assert rs2.get_state()[2] == st[2]


@pytest.mark.gpu
@pytest.mark.parametrize("shape,draws_before", [((96, 96, 96), 0), ((40, 50, 64), 777), ((3, 5, 7), 5), ((1, 2, 624), 0)])
def test_exact_noise_parallel_path_is_numpy_bit_for_bit(shape, draws_before):
    """AdditiveGaussianNoise's default (exact) path at sizes where the word stream is cut into up to 256 chunks whose keys come from GF(2) jump-ahead on the device:
    the field equals numpy's `m + RandomState.normal(0, std, size)` bit for bit in float32, consecutive calls continue the stream (gauss cache across calls), and the
    host RandomState ends in numpy's state (key, position, cache) - also against the round-2 single-workgroup path"""
    from mdeical_image_segmentation_amd.augment.unet3d_augment import transforms as tr
    v = np.random.RandomState(1).rand(*shape).astype(np.float32)
    seeds = {}
    for arm in (True, "serial"):
        rs = np.random.RandomState(99)
        ref = np.random.RandomState(99)
        if draws_before:
            rs.randint(0, 2 ** 32, size=draws_before, dtype=np.uint32)
            ref.randint(0, 2 ** 32, size=draws_before, dtype=np.uint32)
        t = tr.AdditiveGaussianNoise(rs, scale=(0.1, 0.4), execution_probability=1.0, exact=arm)
        outs = []
        for call in range(2):
            got = t(torch.from_numpy(v).cuda()).cpu().numpy()
            assert ref.uniform() < 1.0
            std = ref.uniform(0.1, 0.4)
            want = (v + ref.normal(0, std, size=v.shape)).astype(np.float32)
            assert np.array_equal(got, want), (arm, call, np.abs(got - want).max())
            outs.append(got)
        a, b = rs.get_state(), ref.get_state()
        assert np.array_equal(a[1], b[1]) and a[2] == b[2] and a[3] == b[3] and (a[4] == b[4] or not a[3]), arm
        seeds[arm] = outs
    assert all(np.array_equal(x, y) for x, y in zip(seeds[True], seeds["serial"]))


@pytest.mark.gpu
def test_exact_noise_deferred_state_keeps_the_stream():
    """defer_state=True (what `Transformer` sets for the RandomState it creates per transform): the host RandomState is brought up to date at the start of the NEXT
    call instead of inside the call - the sequence of fields over several calls (execution probability < 1, so that skipped calls interleave) is still numpy's"""
    from mdeical_image_segmentation_amd.augment.unet3d_augment import transforms as tr
    v = np.random.RandomState(2).rand(24, 40, 56).astype(np.float32)
    rs, ref = np.random.RandomState(321), np.random.RandomState(321)
    t = tr.AdditiveGaussianNoise(rs, scale=(0.0, 0.3), execution_probability=0.6, defer_state=True)
    x = torch.from_numpy(v).cuda()
    for call in range(6):
        got = t(x).cpu().numpy()
        want = v
        if ref.uniform() < 0.6:
            std = ref.uniform(0.0, 0.3)
            want = (v + ref.normal(0, std, size=v.shape)).astype(np.float32)
        assert np.array_equal(got, want), call
    t.flush()
    a, b = rs.get_state(), ref.get_state()
    assert np.array_equal(a[1], b[1]) and a[2] == b[2] and a[3] == b[3]
    tf = tr.Transformer({"raw": [{"name": "AdditiveGaussianNoise", "execution_probability": 1.0}]}, {})
    assert tf.raw_transform().transforms[0].defer_state is True
