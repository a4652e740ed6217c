import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


def load_golden(name):
    import numpy as np
    return np.load(os.path.join(GOLDEN, name), allow_pickle=False)


@pytest.fixture
def switches():
    """set kernel-selection switches of libmisamd for one test (mis_dispatch_override, include/misamd.h); everything is reset afterwards"""
    from mdeical_image_segmentation_amd import ops

    def set_(name, value=1):
        ops.dispatch_override(name, value)

    yield set_
    ops.dispatch_override(None, 0)
