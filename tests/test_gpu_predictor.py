"""Patch-tiled 3-D prediction on the device (model/unet3d/predictor.py mirror + csrc/predictor.hip) against the golden replay of the
reference's predictor loop (tests/golden/g8_predictor.npz) and, for the index kernels alone, against numpy."""
import numpy as np
import pytest
import torch

from conftest import load_golden

pytestmark = pytest.mark.gpu


def test_gather_accumulate_finalize_kernels_vs_numpy():
    import ctypes as C
    from mdeical_image_segmentation_amd._lib import check, load, stream_ptr
    from oracle import predictor_oracle as po
    lib = load()
    rng = np.random.RandomState(0)
    raw = rng.randn(9, 14, 11).astype(np.float32)
    patch, stride, halo = (4, 6, 5), (3, 4, 4), (2, 3, 1)
    slices = po.build_slices(raw.shape, patch, stride)
    origins = [[s.start for s in idx] for idx in slices]
    vol = torch.from_numpy(raw).cuda()
    PD, PH, PW = (p + 2 * h for p, h in zip(patch, halo))
    org = torch.tensor(origins, dtype=torch.int32, device="cuda")
    patches = torch.empty(len(origins), 1, PD, PH, PW, device="cuda")
    check(lib.mis_patch_gather_reflect(vol.data_ptr(), 1, 9, 14, 11, org.data_ptr(), len(origins), PD, PH, PW, *halo, patches.data_ptr(), stream_ptr()), "g")
    padded = np.pad(raw, [(h, h) for h in halo], mode="reflect")
    for i, idx in enumerate(slices):
        pidx = tuple(slice(s.start, s.stop + 2 * h) for s, h in zip(idx, halo))
        assert np.array_equal(patches[i, 0].cpu().numpy(), padded[pidx]), i
    # a "model" that returns three fixed channel maps of its input: exercises accumulate + finalize exactly (sums of the same floats)
    def model_fn(x):
        return np.concatenate([x * 0.5, -x, x * x], axis=1).astype(np.float32)
    ref = po.predict_volume(model_fn, raw, patch, stride, halo, 3)
    pmap = torch.zeros(3, 9, 14, 11, device="cuda")
    norm = torch.zeros(9, 14, 11, dtype=torch.uint8, device="cuda")
    for i, (oz, oy, ox) in enumerate(origins):
        pred = torch.from_numpy(model_fn(patches[i:i + 1].cpu().numpy())[0]).cuda().contiguous()
        check(lib.mis_patch_accumulate(pred.data_ptr(), 3, PD, PH, PW, *halo, 0, -1, oz, oy, ox, pmap.data_ptr(), norm.data_ptr(), 9, 14, 11, stream_ptr()), "a")
    prob = torch.empty_like(pmap)
    seg = torch.empty(9, 14, 11, dtype=torch.uint16, device="cuda")
    check(lib.mis_pred_finalize(pmap.data_ptr(), norm.data_ptr(), 3, 9 * 14 * 11, prob.data_ptr(), seg.data_ptr(), stream_ptr()), "f")
    assert np.array_equal(prob.cpu().numpy(), ref)
    assert np.array_equal(seg.cpu().view(torch.int16).numpy().astype(np.uint16), np.argmax(ref, axis=0).astype(np.uint16))


def test_predict_volume_matches_reference_replay():
    from mdeical_image_segmentation_amd.model.unet3d.model import UNet3D
    from mdeical_image_segmentation_amd.model.unet3d.predictor import StandardPredictor, build_origins
    g = load_golden("g8_predictor.npz")
    patch, stride, halo = tuple(int(v) for v in g["patch"]), tuple(int(v) for v in g["stride"]), tuple(int(v) for v in g["halo"])
    assert [list(o) for o in build_origins(g["raw"].shape, patch, stride)] == g["origins"].tolist()
    torch.manual_seed(0)
    net = UNet3D(1, 3, f_maps=[64, 128], num_levels=2).cuda()
    prob = StandardPredictor(net, out_channels=3).predict_volume(g["raw"], patch, stride, halo, batch_size=5)
    flat = prob.cpu().numpy().reshape(-1)
    d = np.abs(flat[g["sample_idx"]] - g["sample"]).max()
    assert d < 1e-4, d
    assert abs(flat.astype(np.float64).sum() - g["stats"][0]) < 1e-4 * g["stats"][1]
    seg = StandardPredictor(net, out_channels=3, save_segmentation=True).predict_volume(g["raw"], patch, stride, halo, batch_size=3)
    seg = seg.cpu().view(torch.int16).numpy().astype(np.uint16)
    res = prob.cpu().numpy()
    top2 = np.sort(res, axis=0)[-2:]
    near = (top2[1] - top2[0]) < 1e-4
    assert np.array_equal(seg[~near], g["seg"][~near]), f"{int((seg != g['seg'])[~near].sum())} arg-max flips away from near-ties"
    assert np.array_equal(seg, np.argmax(res, axis=0).astype(np.uint16))
    one = StandardPredictor(net, out_channels=3, prediction_channel=1).predict_volume(g["raw"], patch, stride, halo)
    assert tuple(one.shape) == (1,) + g["raw"].shape and np.allclose(one.cpu().numpy()[0], res[1], atol=1e-6)
    with pytest.raises(NotImplementedError):
        StandardPredictor(net, out_channels=3)(None)
