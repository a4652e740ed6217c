"""Golden vectors of the REAL reference augmentation classes (build container only; scipy does the rotation).

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_augment.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from _ref_import import import_reference  # noqa: E402


def main():
    ns = import_reference()
    tr = ns.transforms
    rng = np.random.RandomState(5)
    D, H, W = 8, 12, 12
    z, y, x = np.meshgrid(np.arange(D), np.arange(H), np.arange(W), indexing="ij")
    raw = ((z * 0.11 + y * 0.05 - x * 0.07) / 2 + 0.1 * rng.randn(D, H, W)).astype(np.float32)
    raw = np.clip(raw, -1, 1)
    label = (rng.rand(D, H, W) * 4).astype(np.int64)
    out = {"raw": raw, "label": label}
    # (1) the Transformer exactly as the reference's datasets build it: same seed for raw and label pipelines
    axes = [[2, 1]]
    cfg = {"raw": [{"name": "RandomFlip"}, {"name": "RandomRotate90"},
                   {"name": "RandomRotate", "axes": axes, "angle_spectrum": 30, "mode": "reflect", "order": 0},
                   {"name": "RandomContrast", "execution_probability": 0.6}],
           "label": [{"name": "RandomFlip"}, {"name": "RandomRotate90"},
                     {"name": "RandomRotate", "axes": axes, "angle_spectrum": 30, "mode": "reflect", "order": 0}]}
    t = tr.Transformer(cfg, {"mean": 0.05, "std": 1.0})
    out["transformer_seed"] = np.array(t.seed)
    rt, lt = t.raw_transform(), t.label_transform()
    for i in range(6):
        out[f"pipe_raw_{i}"] = np.ascontiguousarray(rt(raw))
        out[f"pipe_label_{i}"] = np.ascontiguousarray(lt(label))
    # (2) single transforms with explicit seeds / all three rotation planes (non-cubic volume)
    D2, H2, W2 = 6, 10, 14
    v = rng.rand(D2, H2, W2).astype(np.float32) * 2 - 1
    out["v"] = v
    for s in range(8):
        rr = tr.RandomRotate(np.random.RandomState(100 + s), angle_spectrum=30, mode="reflect", order=0)
        out[f"rot_{s}"] = np.ascontiguousarray(rr(v))
    for s in range(4):
        out[f"flip_{s}"] = np.ascontiguousarray(tr.RandomFlip(np.random.RandomState(200 + s))(v))
    sq = rng.rand(5, 9, 9).astype(np.float32)
    out["sq"] = sq
    for s in range(6):
        out[f"rot90_{s}"] = np.ascontiguousarray(tr.RandomRotate90(np.random.RandomState(300 + s))(sq))
    c4 = rng.rand(2, 4, 6, 6).astype(np.float32)          # 4-D (C, D, H, W): same op on every channel
    out["c4"] = c4
    out["c4_flip"] = np.ascontiguousarray(tr.RandomFlip(np.random.RandomState(7))(c4))
    out["c4_rot"] = np.ascontiguousarray(tr.RandomRotate(np.random.RandomState(8), axes=[(2, 1)])(c4))
    out["std_auto"] = tr.Standardize()(v)
    out["std_fixed"] = tr.Standardize(mean=0.1, std=0.5)(v)
    out["contrast"] = tr.RandomContrast(np.random.RandomState(9), mean=0.05, execution_probability=1.0)(v)
    out["norm"] = tr.Normalize(min_value=-1.0, max_value=1.0)(v)
    # (3) cubic-spline rotation (order=3: the raw-volume setting of the reference's 3-D configs), all planes + 4-D input
    for s in range(6):
        rr = tr.RandomRotate(np.random.RandomState(400 + s), angle_spectrum=30, mode="reflect", order=3)
        out[f"rot3_{s}"] = np.ascontiguousarray(rr(v))
    out["c4_rot3"] = np.ascontiguousarray(tr.RandomRotate(np.random.RandomState(8), axes=[(2, 1)], order=3)(c4))
    # (4) elastic deformation (order 3 on raw, order 0 on labels, same seed -> same fields); small sigma so the field varies over the volume
    ev = rng.rand(10, 12, 14).astype(np.float32)
    el = (rng.rand(10, 12, 14) * 5).astype(np.int64)
    out["el_raw"], out["el_label"] = ev, el
    out["el_raw_out"] = np.ascontiguousarray(tr.ElasticDeformation(np.random.RandomState(500), spline_order=3, alpha=15, sigma=3, execution_probability=1.0)(ev))
    out["el_label_out"] = np.ascontiguousarray(tr.ElasticDeformation(np.random.RandomState(500), spline_order=0, alpha=15, sigma=3, execution_probability=1.0)(el))
    out["el_raw_2d"] = np.ascontiguousarray(tr.ElasticDeformation(np.random.RandomState(501), spline_order=3, alpha=2000, sigma=50, execution_probability=1.0,
                                                                   apply_3d=False)(ev))
    # (5) CropToFixed (window, mirror-padded, mixed, centred; 3-D and 4-D; int64 labels) and Poisson noise
    for s_, (size, cen, src) in enumerate([((8, 9), False, "v"), ((13, 20), False, "v"), ((7, 19), False, "v"), ((6, 6), True, "v"), ((12, 31), True, "v"),
                                           ((4, 9), False, "c4"), ((8, 8), False, "label")]):
        out[f"crop_{s_}"] = np.ascontiguousarray(tr.CropToFixed(np.random.RandomState(600 + s_), size=size, centered=cen)(out[src]))
    out["pnorm"] = tr.PercentileNormalizer()(v)
    out["pnorm_5_90"] = tr.PercentileNormalizer(pmin=5, pmax=90)(v)
    out["poisson"] = tr.AdditivePoissonNoise(np.random.RandomState(700), lam=(0.5, 3.0), execution_probability=1.0)(v)
    np.savez_compressed(os.path.join(HERE, "g5_augment.npz"), **out)
    print("wrote g5_augment.npz", sum(a.nbytes for a in out.values()) // 1024, "KiB; seed", t.seed)


if __name__ == "__main__":
    main()
