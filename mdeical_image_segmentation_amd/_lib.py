"""ctypes binding of libmisamd.so (include/misamd.h).

The shared library is the product; this module only marshals raw device pointers and sizes.
There is NO fallback: if the library is missing or a call fails, an exception is raised.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# MISAMD_LIB: load another build of the library (diagnostic builds of scripts/ppt_ablate.sh, linked to a scratch path so that they never replace the shipped one)
LIB_PATH = os.environ.get("MISAMD_LIB") or os.path.join(_HERE, "libmisamd.so")

MIS_F32, MIS_BF16 = 0, 1
OUT_PLAIN, OUT_SHUFFLE2, OUT_UNSHUFFLE2 = 0, 1, 2

EXPORTS = [
    "mis_last_error", "mis_version", "mis_abi_struct_count", "mis_abi_struct_name", "mis_abi_layout", "mis_conv_igemm", "mis_conv_stats_rows", "mis_conv_stats_reduce", "mis_conv_stats_reduce_workspace_bytes", "mis_wgrad_workspace_bytes", "mis_wgrad", "mis_wgrad_reduce_batch",
    "mis_conv_last_dispatch", "mis_wgrad_last_dispatch", "mis_wgrad_last_nsplit", "mis_dispatch_override", "mis_dispatch_switch", "mis_gn_apply",
    "mis_mt19937_words", "mis_legacy_normal", "mis_mt_jump", "mis_mt_generate", "mis_legacy_normal_par_workspace_bytes", "mis_legacy_normal_par",
    "mis_comm_unique_id", "mis_comm_init", "mis_comm_world", "mis_allreduce_bucket", "mis_comm_finalize",
    "mis_conv3x3_first_fwd", "mis_conv3x3_first_fwd_rb", "mis_relu_bits_bytes", "mis_relu_bits", "mis_conv3x3_first_wgrad_workspace_bytes", "mis_conv3x3_first_wgrad",
    "mis_colsum_workspace_bytes", "mis_colsum", "mis_maxpool2_fwd", "mis_maxpool2_bwd", "mis_maxpool2_fwd_pb", "mis_maxpool2_bwd_pb",
    "mis_pack_conv_weight", "mis_pack_convt_weight", "mis_pack_batch", "mis_pack_batch2", "mis_head_workspace_bytes", "mis_head_loss", "mis_conv3x3_head_fused_eligible", "mis_conv3x3_head_fused",
    "mis_adamw_workspace_bytes", "mis_sumsq", "mis_adamw_step", "mis_adamw_step_dev", "mis_sumsq_npartials",
    "mis_chanstats_workspace_bytes", "mis_chanstats", "mis_nchw_to_nhwc", "mis_nhwc_to_nchw", "mis_probe_mfma",
    "mis_gn_fwd_finalize", "mis_gn_bwd_stats_workspace_bytes", "mis_gn_bwd_stats", "mis_gn_bwd_stats_from_dw_workspace_bytes", "mis_gn_bwd_stats_from_dw", "mis_gn_cond", "mis_gn_bwd_finalize", "mis_gn_bwd_apply",
    "mis_first3d_fwd", "mis_first3d_bwd_workspace_bytes", "mis_first3d_bwd", "mis_relu_mask",
    "mis_convt3_col2im", "mis_convt3_im2col", "mis_seg_metrics_workspace_bytes", "mis_seg_metrics", "mis_iou3d_counts", "mis_se_fc_fwd", "mis_se_apply_fwd", "mis_se_bwd_workspace_bytes", "mis_se_bwd_reduce", "mis_se_fc_bwd", "mis_se_bwd_apply", "mis_debug_tile_queue", "mis_debug_tile_queue_poke", "mis_tile_queue_init", "mis_tile_queue_reset", "mis_tile_queue_errors", "mis_build_has_experiments", "mis_se_layer_fwd", "mis_se_layer_bwd_reduce", "mis_se_layer_bwd_apply", "mis_patch_gather_reflect", "mis_patch_accumulate", "mis_pred_finalize", "mis_bcedice_workspace_bytes", "mis_bcedice_fwd", "mis_bcedice_bwd", "mis_loss_workspace_bytes", "mis_ce3d_fwd", "mis_ce3d_bwd", "mis_pointloss_fwd", "mis_pointloss_bwd", "mis_maxpoolk_fwd", "mis_maxpoolk_bwd", "mis_bilinear_up_fwd", "mis_bilinear_up_bwd_workspace_bytes", "mis_bilinear_up_bwd", "mis_upconv_gather_fwd_workspace_bytes", "mis_upconv_gather_fwd", "mis_upconv_gather_bwd", "mis_cgm_gate", "mis_scale_sigmoid", "mis_segloss_workspace_bytes", "mis_segloss_fwd", "mis_segloss_bwd", "mis_add_act", "mis_expand1_fwd", "mis_expand1_bwd_workspace_bytes", "mis_expand1_bwd", "mis_bn_fwd_finalize", "mis_bn_bwd_finalize", "mis_affine_act", "mis_bn_bwd_stats_workspace_bytes", "mis_bn_bwd_stats", "mis_bn_bwd_apply",
    "mis_norm_act_fwd", "mis_norm_act_bwd", "mis_mask_scale", "mis_gn_fwd_finalize_ld", "mis_gn_bwd_finalize_ld", "mis_pool3d_fwd", "mis_pool3d_bwd", "mis_gather3d_fwd", "mis_gather3d_bwd",
    "mis_aug2d_u8", "mis_aug_flip_rot90", "mis_aug_crop_reflect", "mis_aug_rotate0", "mis_aug_rotate0_mode", "mis_aug_rotate3_workspace_bytes", "mis_aug_rotate3", "mis_aug_rotate_spline", "mis_aug_gauss1d", "mis_aug_gauss1d_f32", "mis_aug_map_coordinates", "mis_aug_pointwise", "mis_aug_contrast", "mis_minmax",
]


class MisError(RuntimeError):
    pass


class ConvDesc(C.Structure):
    _fields_ = [
        ("dtype", C.c_int), ("ksize", C.c_int),
        ("N", C.c_int), ("D", C.c_int), ("H", C.c_int), ("W", C.c_int), ("is3d", C.c_int),
        ("Cin", C.c_int), ("Cout", C.c_int),
        ("x0", C.c_void_p), ("x0_ld", C.c_int), ("x0_D", C.c_int), ("x0_H", C.c_int), ("x0_W", C.c_int),
        ("x1", C.c_void_p), ("x1_ld", C.c_int), ("x1_D", C.c_int), ("x1_H", C.c_int), ("x1_W", C.c_int),
        ("Cin0", C.c_int),
        ("in_scale", C.c_void_p), ("in_shift", C.c_void_p),
        ("w", C.c_void_p), ("bias", C.c_void_p), ("relu", C.c_int),
        ("mask", C.c_void_p), ("mask_ld", C.c_int),
        ("y0", C.c_void_p), ("y0_ld", C.c_int), ("y0_mode", C.c_int),
        ("y1", C.c_void_p), ("y1_ld", C.c_int), ("y1_mode", C.c_int),
        ("Cout0", C.c_int),
        ("relu_bits", C.c_void_p), ("mask_bits", C.c_void_p),
        ("gn_p", C.c_void_p), ("gn_q", C.c_void_p), ("gn_r", C.c_void_p), ("gn_ld", C.c_int), ("gn_relu", C.c_int),
        ("st_mode", C.c_int), ("st_x0", C.c_void_p), ("st_x0_ld", C.c_int), ("st_x1", C.c_void_p), ("st_x1_ld", C.c_int), ("st_c0", C.c_int), ("st_up", C.c_int),
        ("st_part", C.c_void_p),
    ]


class WgradDesc(C.Structure):
    _fields_ = [
        ("dtype", C.c_int), ("ksize", C.c_int),
        ("N", C.c_int), ("D", C.c_int), ("H", C.c_int), ("W", C.c_int), ("is3d", C.c_int),
        ("Cin", C.c_int), ("Cout", C.c_int),
        ("x0", C.c_void_p), ("x0_ld", C.c_int), ("x0_D", C.c_int), ("x0_H", C.c_int), ("x0_W", C.c_int),
        ("x1", C.c_void_p), ("x1_ld", C.c_int), ("x1_D", C.c_int), ("x1_H", C.c_int), ("x1_W", C.c_int),
        ("Cin0", C.c_int),
        ("in_scale", C.c_void_p), ("in_shift", C.c_void_p),
        ("dy", C.c_void_p), ("dy_ld", C.c_int),
        ("workspace", C.c_void_p), ("workspace_bytes", C.c_size_t),
        ("dw", C.c_void_p), ("dw_layout", C.c_int), ("alpha", C.c_float),
        ("dbias", C.c_void_p),
        ("reduce_stream", C.c_void_p),
        ("dw_per_sample", C.c_void_p), ("dbias_per_sample", C.c_void_p),
        ("defer", C.c_void_p),
    ]


class WgradReduceItem(C.Structure):
    _fields_ = [
        ("partial", C.c_void_p), ("dw", C.c_void_p), ("bias_partial", C.c_void_p), ("dbias", C.c_void_p),
        ("nsplit", C.c_int), ("TT", C.c_int), ("Cin", C.c_int), ("Cout", C.c_int), ("dw_layout", C.c_int), ("alpha", C.c_float),
    ]


class HeadDesc(C.Structure):
    _fields_ = [
        ("dtype", C.c_int), ("loss", C.c_int),
        ("npix_per_image", C.c_longlong), ("N", C.c_int), ("Cfeat", C.c_int), ("C", C.c_int),
        ("y", C.c_void_p), ("y_ld", C.c_int),
        ("w", C.c_void_p), ("b", C.c_void_p),
        ("labels", C.c_void_p),
        ("logits", C.c_void_p), ("argmax", C.c_void_p),
        ("workspace", C.c_void_p), ("workspace_bytes", C.c_size_t),
        ("loss_out", C.c_void_p),
        ("dy", C.c_void_p), ("dy_ld", C.c_int),
        ("dw", C.c_void_p), ("db", C.c_void_p),
        ("grad_scale", C.c_float), ("alpha", C.c_float), ("beta", C.c_float), ("phase", C.c_int),
    ]


class PackItem(C.Structure):
    _fields_ = [("w", C.c_void_p), ("w_fwd", C.c_void_p), ("w_dgrad", C.c_void_p), ("rows", C.c_int), ("cols", C.c_int), ("taps", C.c_int), ("kind", C.c_int)]


class PackItem2(C.Structure):
    _fields_ = [("w", C.c_void_p), ("w_fwd", C.c_void_p), ("w_dgrad", C.c_void_p), ("rows", C.c_int), ("cols", C.c_int), ("taps", C.c_int), ("kind", C.c_int),
                ("blk0", C.c_int), ("nbx", C.c_int)]


# C struct (include/misamd.h) -> its Python mirror.  abi_mismatches() compares every mirror with the layout the LOADED library reports (mis_abi_layout); load() refuses
# a library whose structs differ from these mirrors, and tests/test_cabi_symbols.py shows that a swapped / missing field is caught.
ABI_MIRRORS = {"MisConvDesc": ConvDesc, "MisWgradDesc": WgradDesc, "MisWgradReduceItem": WgradReduceItem, "MisHeadDesc": HeadDesc, "MisPackItem": PackItem,
               "MisPackItem2": PackItem2}


def pack_item2_dtype():
    """numpy record dtype of MisPackItem2 (the device-resident table of mis_pack_batch2), derived from the ctypes mirror that abi_mismatches() checks"""
    import numpy as np
    code = {C.c_void_p: "<u8", C.c_int: "<i4", C.c_float: "<f4", C.c_longlong: "<i8", C.c_size_t: "<u8"}
    fields = PackItem2._fields_
    return np.dtype({"names": [n for n, _ in fields], "formats": [code[t] for _, t in fields], "offsets": [getattr(PackItem2, n).offset for n, _ in fields],
                     "itemsize": C.sizeof(PackItem2)})


def abi_mismatches(lib, mirrors=None):
    """list of differences between the ctypes mirrors and the structs of the loaded library (empty = they agree): size, field count, and per field name, offset, size"""
    mirrors = ABI_MIRRORS if mirrors is None else mirrors
    out = []
    covered = {lib.mis_abi_struct_name(i).decode() for i in range(lib.mis_abi_struct_count())}
    for missing in sorted(covered - set(mirrors)):
        out.append(f"{missing}: the library describes it, no Python mirror is registered")
    for cname, cls in mirrors.items():
        size = C.c_size_t(0)
        cap = 128
        names, offs, sizes = (C.c_char_p * cap)(), (C.c_size_t * cap)(), (C.c_size_t * cap)()
        n = lib.mis_abi_layout(cname.encode(), C.byref(size), names, offs, sizes, cap)
        if n < 0:
            out.append(f"{cname}: unknown to the library")
            continue
        if size.value != C.sizeof(cls):
            out.append(f"{cname}: sizeof {size.value} in C, {C.sizeof(cls)} in {cls.__name__}")
        fields = cls._fields_
        if n != len(fields):
            out.append(f"{cname}: {n} fields in C, {len(fields)} in {cls.__name__}")
        for i in range(min(n, len(fields))):
            fname = fields[i][0]
            desc = getattr(cls, fname)
            if names[i].decode() != fname or offs[i] != desc.offset or sizes[i] != desc.size:
                out.append(f"{cname} field {i}: C has {names[i].decode()} @ {offs[i]} ({sizes[i]} B), {cls.__name__} has {fname} @ {desc.offset} ({desc.size} B)")
    return out


_lib = None


def load():
    """Load libmisamd.so (no GPU needed to load it). Raises MisError when it is missing."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise MisError(f"{LIB_PATH} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                       "(make -C mdeical_image_segmentation_amd/csrc). There is no CPU fallback.")
    # torch ships its own HIP runtime (torch/lib/libamdhip64.so, SONAME libamdhip64.so.7); it must be in the process
    # BEFORE libmisamd.so is loaded so that both share ONE runtime (device pointers, streams).  Loading ours first
    # binds it to /opt/rocm's copy and its launches then fail with "no ROCm-capable device is detected".
    import torch  # noqa: F401
    tl = os.path.join(os.path.dirname(torch.__file__), "lib", "libamdhip64.so")
    if os.path.exists(tl):
        C.CDLL(tl, mode=C.RTLD_GLOBAL)
    lib = C.CDLL(LIB_PATH)
    lib.mis_last_error.restype = C.c_char_p
    lib.mis_last_error.argtypes = []
    lib.mis_version.restype = C.c_int
    lib.mis_version.argtypes = []
    if not os.environ.get("MISAMD_LIB"):                 # (an older build loaded for an A/B has no layout entry points)
        lib.mis_abi_struct_count.restype = C.c_int
        lib.mis_abi_struct_count.argtypes = []
        lib.mis_abi_struct_name.restype = C.c_char_p
        lib.mis_abi_struct_name.argtypes = [C.c_int]
        lib.mis_abi_layout.restype = C.c_int
        lib.mis_abi_layout.argtypes = [C.c_char_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
        bad = abi_mismatches(lib)
        if bad:
            raise MisError("the ctypes mirrors in _lib.py do not match the structs of " + LIB_PATH + " (rebuild, or update the mirror): " + "; ".join(bad))
    for name in ("mis_conv_last_dispatch", "mis_wgrad_last_dispatch"):
        getattr(lib, name).restype = C.c_char_p
        getattr(lib, name).argtypes = []
    lib.mis_wgrad_last_nsplit.restype = C.c_int
    lib.mis_wgrad_last_nsplit.argtypes = []
    lib.mis_dispatch_override.restype = C.c_int
    lib.mis_dispatch_override.argtypes = [C.c_char_p, C.c_int]
    lib.mis_dispatch_switch.restype = C.c_int
    lib.mis_dispatch_switch.argtypes = [C.c_char_p]
    if hasattr(lib, "mis_wgrad_reduce_batch"):          # (absent from an older build loaded through MISAMD_LIB for an A/B: scripts/ab_prev_lib.sh sets MISAMD_REDUCE_PER_LAYER=1 for it)
        lib.mis_wgrad_reduce_batch.restype = C.c_int
        lib.mis_wgrad_reduce_batch.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
    lib.mis_comm_unique_id.argtypes = [C.c_void_p]
    lib.mis_comm_init.argtypes = [C.c_void_p, C.c_int, C.c_int]
    lib.mis_comm_world.argtypes = []
    lib.mis_allreduce_bucket.argtypes = [C.c_void_p, C.c_longlong, C.c_void_p]
    lib.mis_comm_finalize.argtypes = []
    for name in ("mis_comm_unique_id", "mis_comm_init", "mis_comm_world", "mis_allreduce_bucket", "mis_comm_finalize"):
        getattr(lib, name).restype = C.c_int
    for name in ("mis_wgrad_workspace_bytes", "mis_head_workspace_bytes"):
        getattr(lib, name).restype = C.c_size_t
        getattr(lib, name).argtypes = [C.c_void_p]
    lib.mis_conv_stats_reduce_workspace_bytes.restype = C.c_size_t
    lib.mis_conv_stats_reduce_workspace_bytes.argtypes = [C.c_int, C.c_int]
    lib.mis_conv_stats_rows.restype = C.c_longlong
    lib.mis_conv_stats_rows.argtypes = [C.c_void_p]
    lib.mis_relu_bits_bytes.restype = C.c_size_t
    lib.mis_relu_bits_bytes.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int]
    lib.mis_conv3x3_first_wgrad_workspace_bytes.restype = C.c_size_t
    lib.mis_conv3x3_first_wgrad_workspace_bytes.argtypes = [C.c_int] * 5
    lib.mis_colsum_workspace_bytes.restype = C.c_size_t
    lib.mis_colsum_workspace_bytes.argtypes = [C.c_longlong, C.c_int]
    lib.mis_adamw_workspace_bytes.restype = C.c_size_t
    lib.mis_adamw_workspace_bytes.argtypes = [C.c_longlong]
    lib.mis_chanstats_workspace_bytes.restype = C.c_size_t
    lib.mis_chanstats_workspace_bytes.argtypes = [C.c_int, C.c_longlong, C.c_int]
    lib.mis_sumsq_npartials.restype = C.c_int
    lib.mis_sumsq_npartials.argtypes = [C.c_longlong]

    lib.mis_gn_bwd_stats_workspace_bytes.restype = C.c_size_t
    lib.mis_gn_bwd_stats_workspace_bytes.argtypes = [C.c_int, C.c_int]
    lib.mis_gn_bwd_stats_from_dw_workspace_bytes.restype = C.c_size_t
    lib.mis_gn_bwd_stats_from_dw_workspace_bytes.argtypes = [C.c_int, C.c_int]
    lib.mis_first3d_bwd_workspace_bytes.restype = C.c_size_t
    lib.mis_first3d_bwd_workspace_bytes.argtypes = []
    lib.mis_bilinear_up_bwd_workspace_bytes.restype = C.c_size_t
    lib.mis_bn_bwd_stats_workspace_bytes.restype = C.c_size_t
    lib.mis_loss_workspace_bytes.restype = C.c_size_t
    lib.mis_upconv_gather_fwd_workspace_bytes.restype = C.c_size_t
    lib.mis_upconv_gather_fwd_workspace_bytes.argtypes = [C.c_int] * 6
    lib.mis_loss_workspace_bytes.argtypes = []
    lib.mis_se_bwd_workspace_bytes.restype = C.c_size_t
    lib.mis_se_bwd_workspace_bytes.argtypes = [C.c_int] * 2
    lib.mis_bn_bwd_stats_workspace_bytes.argtypes = [C.c_int] * 2
    lib.mis_bilinear_up_bwd_workspace_bytes.argtypes = [C.c_int] * 5
    lib.mis_segloss_workspace_bytes.restype = C.c_size_t
    lib.mis_segloss_workspace_bytes.argtypes = [C.c_int] * 3
    lib.mis_expand1_bwd_workspace_bytes.restype = C.c_size_t
    lib.mis_expand1_bwd_workspace_bytes.argtypes = [C.c_int]
    lib.mis_bcedice_workspace_bytes.restype = C.c_size_t
    lib.mis_bcedice_workspace_bytes.argtypes = [C.c_int]
    lib.mis_seg_metrics_workspace_bytes.restype = C.c_size_t
    lib.mis_seg_metrics_workspace_bytes.argtypes = [C.c_int, C.c_longlong]
    lib.mis_legacy_normal_par_workspace_bytes.restype = C.c_size_t
    lib.mis_legacy_normal_par_workspace_bytes.argtypes = [C.c_longlong]
    lib.mis_aug_rotate3_workspace_bytes.restype = C.c_size_t
    lib.mis_aug_rotate3_workspace_bytes.argtypes = [C.c_longlong, C.c_int, C.c_int, C.c_int]
    vp, i, ll, f, dbl = C.c_void_p, C.c_int, C.c_longlong, C.c_float, C.c_double
    sigs = {
        "mis_conv_igemm": [vp, vp],
        "mis_wgrad": [vp, vp],
        "mis_conv3x3_first_fwd": [i, vp, i, i, i, i, vp, vp, vp, i, i, vp],
        "mis_conv3x3_first_fwd_rb": [i, vp, i, i, i, i, vp, vp, vp, i, i, vp, vp],
        "mis_relu_bits": [vp, i, i, i, i, i, vp, vp],
        "mis_conv3x3_first_wgrad": [i, vp, i, i, i, i, vp, i, i, vp, vp, vp, vp],
        "mis_colsum": [i, vp, i, ll, i, i, f, vp, vp, vp],
        "mis_maxpool2_fwd": [i, vp, i, vp, i, i, i, i, i, i, vp],
        "mis_maxpool2_fwd_pb": [i, vp, i, vp, i, i, i, i, i, vp, vp],
        "mis_maxpool2_bwd_pb": [i, vp, vp, i, vp, i, vp, i, i, i, i, i, vp],
        "mis_maxpool2_bwd": [i, vp, i, vp, i, vp, i, vp, i, i, i, i, i, i, i, vp],
        "mis_pack_conv_weight": [i, vp, i, i, i, vp, vp, vp],
        "mis_pack_convt_weight": [i, vp, i, i, vp, vp, vp],
        "mis_head_loss": [vp, vp],
        "mis_sumsq": [vp, ll, vp, vp],
        "mis_adamw_step": [vp, vp, vp, vp, ll, vp, i, f, f, f, f, f, f, i, vp, vp],
        "mis_adamw_step_dev": [vp, vp, vp, vp, ll, vp, i, f, vp, f, f, f, f, vp, i, vp, vp, vp],
        "mis_chanstats": [i, vp, i, i, ll, i, vp, vp, vp, vp],
        "mis_nchw_to_nhwc": [i, vp, vp, i, i, i, ll, vp],
        "mis_nhwc_to_nchw": [i, vp, i, vp, i, i, ll, vp],
        "mis_probe_mfma": [i, vp, vp, vp, vp],
        "mis_gn_fwd_finalize": [vp, vp, i, f, vp, vp, i, f, i, i, dbl, vp, vp, f, i, vp, vp, vp, vp, vp],
        "mis_gn_bwd_stats": [i, vp, i, vp, i, i, i, i, i, i, i, vp, vp, vp, i, i, vp],
        "mis_gn_apply": [i, vp, i, i, i, i, i, i, i, vp, vp, i, i, vp, i, vp],
        "mis_gn_bwd_stats_from_dw": [i, vp, i, i, i, i, i, i, vp, vp, i, vp, vp, vp, i, vp, i, i, vp, vp, vp, vp],
        "mis_gn_bwd_finalize": [vp, vp, vp, vp, vp, i, i, i, dbl, vp, vp, vp, vp, vp, vp],
        "mis_gn_bwd_apply": [i, vp, i, vp, i, i, i, i, i, i, i, vp, vp, vp, i, i, i, vp, i, vp, i, vp],
        "mis_first3d_fwd": [i, vp, vp, vp, i, i, i, i, i, vp, i, vp, i, i, vp],
        "mis_first3d_bwd": [i, vp, vp, vp, vp, vp, i, i, i, i, vp, i, i, vp, i, vp, vp, vp, vp, vp, vp],
        "mis_relu_mask": [i, vp, i, vp, i, vp, i, ll, i, vp],
        "mis_convt3_col2im": [i, vp, vp, i, i, i, i, i, i, vp],
        "mis_convt3_im2col": [i, vp, i, vp, i, i, i, i, i, vp],
        "mis_seg_metrics": [vp, vp, i, ll, i, i, f, vp, vp, vp],
        "mis_iou3d_counts": [vp, vp, i, i, i, ll, i, ll, vp, vp],
        "mis_se_fc_fwd": [vp, C.c_double, vp, vp, vp, vp, i, i, vp, vp, vp, vp],
        "mis_se_apply_fwd": [i, vp, i, i, ll, i, vp, vp, vp, vp, vp, i, vp],
        "mis_se_bwd_reduce": [i, vp, i, vp, i, i, ll, i, vp, vp, vp, vp, vp, vp, vp, vp],
        "mis_se_fc_bwd": [vp, vp, vp, vp, vp, vp, i, i, C.c_double, vp, vp, vp, vp, vp, vp, vp],
        "mis_se_bwd_apply": [i, vp, i, vp, i, i, ll, i, vp, vp, vp, vp, vp, vp, i, vp],
        "mis_debug_tile_queue": [vp, vp],
        "mis_conv_stats_reduce": [vp, i, ll, i, vp, vp, vp, vp],
        "mis_debug_tile_queue_poke": [vp, i, C.c_uint],
        "mis_tile_queue_init": [],
        "mis_tile_queue_reset": [vp],
        "mis_tile_queue_errors": [],
        "mis_build_has_experiments": [],
        "mis_se_layer_fwd": [i, vp, i, i, ll, i, vp, vp, vp, vp, vp, i, i, vp],
        "mis_se_layer_bwd_reduce": [i, vp, i, vp, i, i, ll, i, vp, vp, vp, vp, vp, vp, vp, i, vp],
        "mis_se_layer_bwd_apply": [i, vp, i, vp, i, i, ll, i, vp, vp, vp, vp, vp, vp, i, i, i, vp],
        "mis_patch_gather_reflect": [vp, i, i, i, i, vp, i, i, i, i, i, i, i, vp, vp],
        "mis_patch_accumulate": [vp, i, i, i, i, i, i, i, i, i, i, i, i, vp, vp, i, i, i, vp],
        "mis_pred_finalize": [vp, vp, i, ll, vp, vp, vp],
        "mis_bcedice_fwd": [vp, vp, i, i, ll, f, f, i, vp, vp, vp],
        "mis_bcedice_bwd": [vp, vp, i, i, ll, f, f, i, vp, vp, vp, vp],
        "mis_ce3d_fwd": [vp, vp, i, i, ll, ll, vp, vp, vp],
        "mis_ce3d_bwd": [vp, vp, i, i, ll, ll, vp, vp, vp, vp],
        "mis_pointloss_fwd": [i, vp, vp, ll, vp, vp, vp],
        "mis_pointloss_bwd": [i, vp, vp, ll, vp, vp, vp],
        "mis_maxpoolk_fwd": [i, vp, i, vp, i, i, i, i, i, i, vp],
        "mis_maxpoolk_bwd": [i, vp, i, vp, i, vp, i, i, i, i, i, i, vp],
        "mis_bilinear_up_fwd": [i, vp, i, vp, i, i, i, i, i, i, vp],
        "mis_bilinear_up_bwd": [i, vp, i, vp, i, i, i, i, i, i, vp, vp],
        "mis_bn_bwd_stats": [i, vp, i, vp, i, i, ll, i, vp, vp, vp, vp, vp, vp],
        "mis_bn_bwd_apply": [i, vp, i, vp, i, i, ll, i, vp, vp, vp, vp, vp, vp, i, vp],
        "mis_cgm_gate": [i, vp, i, i, ll, i, vp, vp, vp, vp, vp],
        "mis_scale_sigmoid": [vp, vp, vp, i, ll, vp, vp],
        "mis_upconv_gather_fwd": [i, vp, vp, i, vp, i, i, i, i, i, vp, vp],
        "mis_upconv_gather_bwd": [i, vp, i, vp, i, i, i, i, i, vp],
        "mis_segloss_fwd": [vp, vp, i, i, i, f, f, f, vp, vp, vp],
        "mis_segloss_bwd": [vp, i, i, i, vp, vp, vp, vp, i, vp],
        "mis_add_act": [i, vp, i, vp, i, vp, i, ll, i, i, vp],
        "mis_expand1_fwd": [i, vp, vp, vp, vp, i, ll, i, vp],
        "mis_expand1_bwd": [i, vp, vp, i, ll, i, vp, vp, vp, vp],
        "mis_bn_fwd_finalize": [vp, vp, i, i, dbl, vp, vp, f, f, vp, vp, i, vp, vp, vp, vp, vp],
        "mis_bn_bwd_finalize": [vp, vp, vp, vp, vp, i, i, dbl, i, vp, vp, vp, vp, vp, vp],
        "mis_affine_act": [i, vp, i, vp, i, i, ll, i, vp, vp, i, vp],
        "mis_aug2d_u8": [vp, vp, i, i, i, i, i, i, i, i, i, i, f, f, vp, vp, vp],
        "mis_aug_flip_rot90": [vp, vp, ll, i, i, i, i, i, i, vp],
        "mis_aug_crop_reflect": [vp, vp, ll, i, i, i, i, i, i, i, vp],
        "mis_aug_rotate0": [vp, vp, ll, i, i, i, i, i, C.POINTER(C.c_double), C.POINTER(C.c_double), i, vp],
        "mis_aug_rotate0_mode": [vp, vp, ll, i, i, i, i, i, C.POINTER(C.c_double), C.POINTER(C.c_double), i, i, C.c_ulonglong, vp],
        "mis_aug_rotate3": [vp, vp, vp, ll, i, i, i, i, i, C.POINTER(C.c_double), C.POINTER(C.c_double), vp],
        "mis_aug_rotate_spline": [vp, vp, vp, ll, i, i, i, i, i, C.POINTER(C.c_double), C.POINTER(C.c_double), i, vp],
        "mis_aug_gauss1d": [vp, vp, ll, i, i, i, i, vp, i, vp],
        "mis_aug_gauss1d_f32": [vp, vp, ll, i, i, i, i, vp, i, i, vp],
        "mis_aug_map_coordinates": [vp, vp, vp, ll, i, i, i, vp, vp, vp, dbl, i, i, vp],
        "mis_aug_pointwise": [vp, vp, ll, f, f, i, f, f, f, C.c_ulonglong, vp],
        "mis_aug_contrast": [vp, vp, ll, f, f, vp],
        "mis_minmax": [vp, ll, vp, vp, vp],
        "mis_pack_batch": [i, vp, i, i, i, vp],
        "mis_pack_batch2": [i, vp, i, i, vp],
        "mis_norm_act_fwd": [i, vp, i, vp, i, i, ll, i, vp, vp, i, f, vp],
        "mis_mask_scale": [i, vp, i, vp, i, vp, i, ll, i, f, vp],
        "mis_norm_act_bwd": [i, vp, i, vp, i, vp, i, i, ll, i, vp, vp, i, f, vp],
        "mis_gn_fwd_finalize_ld": [vp, vp, i, i, i, i, dbl, vp, vp, f, vp, vp, vp, vp, vp],
        "mis_gn_bwd_finalize_ld": [vp, vp, vp, vp, vp, i, i, i, i, dbl, vp, vp, vp, vp, vp, vp],
        "mis_pool3d_fwd": [i, i, i, i, i, vp, i, vp, i, i, i, i, i, i, vp],
        "mis_pool3d_bwd": [i, i, i, i, i, vp, i, vp, i, vp, i, i, i, i, i, i, vp],
        "mis_gather3d_fwd": [i, vp, i, vp, i, i, i, i, i, i, i, i, i, vp, vp, vp, vp],
        "mis_gather3d_bwd": [i, vp, i, vp, i, i, i, i, i, i, i, i, i, vp, vp, vp, vp],
        "mis_mt19937_words": [vp, vp, vp, ll, vp],
        "mis_legacy_normal": [vp, ll, vp, vp, ll, dbl, i, dbl, vp, vp],
        "mis_mt_jump": [vp, i, i, i, vp, i, i, i, vp],
        "mis_mt_generate": [vp, i, ll, ll, ll, vp, ll, vp, vp, ll, vp],
        "mis_legacy_normal_par": [vp, ll, vp, vp, ll, dbl, i, dbl, vp, vp, vp],
        "mis_gn_cond": [vp, vp, vp, vp, i, C.c_float, vp, vp],
        "mis_conv3x3_head_fused_eligible": [vp, vp],
        "mis_conv3x3_head_fused": [vp, vp, vp],
    }
    for name, args in sigs.items():
        fn = getattr(lib, name)
        fn.restype = C.c_int
        fn.argtypes = args
    _lib = lib
    return lib


# the sources that define the kernels whose HBM traffic profiles/traffic.json records (the 3x3 convolution / weight-gradient kernels of the benchmark)
TRAFFIC_SOURCES = ("common.hpp", "conv_args.hpp", "conv_pp_common.hpp", "conv_igemm.hip", "conv_pp.hip", "conv_ppd.hip", "gemm1_pp.hip", "wgrad_args.hpp", "wgrad.hip", "wgrad_pp.hip")
# ... and of cfg4's (3-D fp32) dominant kernels: profiles/traffic_3d_f32.json
TRAFFIC_SOURCES_3D_F32 = ("common.hpp", "conv_args.hpp", "conv_pp_common.hpp", "conv3d_f32.hip", "wgrad_args.hpp", "wgrad_f32.hip")


def source_hash(only=None):
    """sha256 (first 16 hex digits) of the kernel sources + the C header this tree was built from.  `only`: restrict to these csrc file names
    (source_hash(TRAFFIC_SOURCES) ties profiles/traffic.json to the sources of the kernels it was measured on, and to nothing else)"""
    import glob
    import hashlib
    h = hashlib.sha256()
    if only is None:
        files = sorted(glob.glob(os.path.join(_HERE, "csrc", "*.hip")) + glob.glob(os.path.join(_HERE, "csrc", "*.hpp")) +
                       glob.glob(os.path.join(_HERE, "csrc", "*.cpp")) + [os.path.join(_HERE, "..", "include", "misamd.h")])
    else:
        files = [os.path.join(_HERE, "csrc", f) for f in sorted(only)]
    for f in files:
        h.update(os.path.basename(f).encode())
        h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


def check(rc, what=""):
    if rc != 0:
        msg = load().mis_last_error().decode("utf-8", "replace")
        raise MisError(f"{what} failed (rc={rc}): {msg}")


def dtype_code(torch_dtype):
    import torch
    if torch_dtype == torch.float32:
        return MIS_F32
    if torch_dtype == torch.bfloat16:
        return MIS_BF16
    raise MisError(f"unsupported dtype {torch_dtype}")


def ptr(t):
    """Device pointer of a torch tensor (None -> NULL)."""
    return None if t is None else t.data_ptr()


_raw_stream = None


def stream_ptr():
    """raw hipStream_t of torch's CURRENT stream on the current device (looked up per launch: it follows torch.cuda.stream(...) contexts);
    the private raw accessor is ~20x cheaper than building a torch.cuda.Stream object for every kernel launch"""
    global _raw_stream
    import torch
    if _raw_stream is None:
        _raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", False)
    if _raw_stream:
        return _raw_stream(torch.cuda.current_device())
    return torch.cuda.current_stream().cuda_stream
