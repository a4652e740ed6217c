"""Golden vectors of the REAL reference `AdditiveGaussianNoise` (build container only): the noise FIELD numpy's legacy RandomState draws.

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_noise.py

Three consecutive calls on ONE RandomState, so that the polar method's cached second value crosses call boundaries (odd element counts), followed by a
uniform() draw: the mirror must leave the generator in the same state.  float64 outputs as the reference returns them (m + float64 noise)."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from _ref_import import import_reference  # noqa: E402


def main():
    tr = import_reference().transforms
    rng = np.random.RandomState(21)
    vols = [rng.rand(3, 5, 7).astype(np.float32) * 2 - 1,          # 105 elements (odd): leaves a cached value
            rng.rand(9, 11, 13).astype(np.float32) * 2 - 1,        # 1287 (odd): starts from the cache, ends with an empty cache
            rng.rand(16, 24, 24).astype(np.float32) * 2 - 1]       # 9216 (even), several 624-word blocks
    out = {}
    rs = np.random.RandomState(777)
    t = tr.AdditiveGaussianNoise(rs, scale=(0.05, 0.3), execution_probability=1.0)
    for i, v in enumerate(vols):
        out[f"in_{i}"] = v
        out[f"out_{i}"] = np.asarray(t(v), dtype=np.float64)
    out["next_uniform"] = np.array([rs.uniform(), rs.uniform()])
    # a call that does not fire (execution_probability 0) must consume exactly one uniform draw
    rs2 = np.random.RandomState(778)
    t2 = tr.AdditiveGaussianNoise(rs2, scale=(0.0, 1.0), execution_probability=0.0)
    out["skip_out"] = np.asarray(t2(vols[0]), dtype=np.float64)
    out["skip_next"] = np.array([rs2.uniform()])
    np.savez_compressed(os.path.join(HERE, "g16_gauss_noise.npz"), **out)
    print({k: (v.shape, v.dtype) for k, v in out.items()})


if __name__ == "__main__":
    main()
