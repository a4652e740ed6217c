"""The C-ABI library loads without a GPU and exports every symbol include/misamd.h declares."""
import ctypes
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    src = open(os.path.join(ROOT, "include", "misamd.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(mis_[a-z0-9_]+)\s*\(", src)))


def test_header_symbols_are_exported():
    from mdeical_image_segmentation_amd import _lib
    lib = _lib.load()
    syms = declared_symbols()
    assert len(syms) >= 20
    for s in syms:
        assert hasattr(lib, s), f"{s} declared in misamd.h but not exported"
    assert sorted(_lib.EXPORTS) == syms, "python binding list and header disagree"
    assert lib.mis_version() >= 1


def test_argument_validation_needs_no_gpu():
    from mdeical_image_segmentation_amd import _lib
    lib = _lib.load()
    d = _lib.ConvDesc()
    d.dtype = 7
    assert lib.mis_conv_igemm(ctypes.byref(d), None) != 0
    assert b"dtype" in lib.mis_last_error()
    assert lib.mis_wgrad_workspace_bytes(ctypes.byref(_lib.WgradDesc())) == 0


def test_every_export_has_a_ctypes_signature():
    """Without `argtypes` ctypes passes a Python int as a 32-bit C int: a device POINTER handed to such an entry point is silently truncated and the kernel faults on the GPU
    (round 4: mis_pack_batch2 was added to EXPORTS without its signature - a memory access fault on the box).  Every exported entry point must carry its argument types."""
    from mdeical_image_segmentation_amd import _lib
    lib = _lib.load()
    missing = [n for n in _lib.EXPORTS if getattr(lib, n).argtypes is None]
    assert not missing, f"no ctypes argtypes for {missing}: add them to _lib.load()"
