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


def test_ctypes_mirrors_match_the_c_structs():
    """the ctypes mirrors of _lib.py are hand-copied from include/misamd.h; the library reports sizeof / field names / offsets / sizes of its own structs
    (mis_abi_layout) and every mirror must agree - and the check must actually SEE a swapped, resized or missing field."""
    import ctypes as C
    from mdeical_image_segmentation_amd import _lib
    lib = _lib.load()
    assert lib.mis_abi_struct_count() >= 6
    assert _lib.abi_mismatches(lib) == []

    def variant(cls, fields):
        return type("Broken" + cls.__name__, (C.Structure,), {"_fields_": fields})

    f = list(_lib.ConvDesc._fields_)
    i, j = [n for n, _ in f].index("gn_q"), [n for n, _ in f].index("gn_r")
    f[i], f[j] = f[j], f[i]                                    # two pointers of equal size swapped: same sizeof, same offsets - only the names differ
    bad = _lib.abi_mismatches(lib, {"MisConvDesc": variant(_lib.ConvDesc, f)})
    assert any("gn_q" in b and "gn_r" in b for b in bad), bad
    f = list(_lib.ConvDesc._fields_)
    k = [n for n, _ in f].index("mask_ld")
    f[k], f[k - 1] = f[k - 1], f[k]                            # an int in front of the pointer it follows: offsets move
    assert _lib.abi_mismatches(lib, {"MisConvDesc": variant(_lib.ConvDesc, f)})
    f = [x for x in _lib.WgradDesc._fields_ if x[0] != "defer"]        # a field the C side has grown
    bad = _lib.abi_mismatches(lib, {"MisWgradDesc": variant(_lib.WgradDesc, f)})
    assert any("fields in C" in b for b in bad) and any("sizeof" in b for b in bad), bad
    f = [(n, C.c_int if n == "npix_per_image" else t) for n, t in _lib.HeadDesc._fields_]      # a narrowed field
    assert _lib.abi_mismatches(lib, {"MisHeadDesc": variant(_lib.HeadDesc, f)})
    # the numpy record of the device-resident pack table is derived from the checked mirror
    dt = _lib.pack_item2_dtype()
    assert dt.itemsize == C.sizeof(_lib.PackItem2) == 48 and dt.fields["blk0"][1] == _lib.PackItem2.blk0.offset
    # a struct the library describes but Python does not mirror is reported too
    assert any("no Python mirror" in b for b in _lib.abi_mismatches(lib, {"MisConvDesc": _lib.ConvDesc}))
