"""Import the reference Python package from /root/reference (build container only).

Used ONLY by tests/golden/make_golden*.py to generate the committed fixtures.
Nothing in tests/, bench.py or the product imports this module at run time:
/root/reference does not exist on the GPU box.

Absent third-party modules are replaced by empty in-process stand-ins exactly as
SURVEY.md §8(c) records (the only function that is really *called* on the hot
path is torchvision's center_crop, which is pure slicing).
"""
import importlib.machinery
import importlib.util
import os
import sys
import types

REF = "/root/reference"


def _stub(name, **attrs):
    m = types.ModuleType(name)
    m.__spec__ = importlib.machinery.ModuleSpec(name, None)
    m.__path__ = []
    for k, v in attrs.items():
        setattr(m, k, v)
    sys.modules[name] = m
    return m


def _center_crop(img, size):
    h, w = int(size[0]), int(size[1])
    H, W = img.shape[-2], img.shape[-1]
    top = int(round((H - h) / 2.0))
    left = int(round((W - w) / 2.0))
    return img[..., top:top + h, left:left + w]


def import_reference():
    """Returns a namespace with the reference modules on the hot path."""
    if not os.path.isdir(REF):
        raise RuntimeError("reference tree not present (this only runs in the build container)")
    sys.dont_write_bytecode = True
    import transformers  # noqa: F401  (must come first)

    if "torchvision" not in sys.modules:
        tv = _stub("torchvision")
        tvt = _stub("torchvision.transforms")
        tvf = _stub("torchvision.transforms.functional", center_crop=_center_crop)
        tv.transforms = tvt
        tvt.functional = tvf
    if "pytorch_msssim" not in sys.modules:
        _stub("pytorch_msssim", MS_SSIM=object, ms_ssim=lambda *a, **k: None, SSIM=object, ssim=None)
    for name in ("h5py", "medpy", "medpy.metric", "skimage", "skimage.measure", "skimage.filters",
                 "skimage.segmentation"):
        if name not in sys.modules:
            _stub(name)
    sys.modules["skimage.filters"].gaussian = None
    sys.modules["skimage.segmentation"].find_boundaries = None
    sys.modules["skimage"].measure = sys.modules["skimage.measure"]
    sys.modules["medpy"].metric = sys.modules["medpy.metric"]
    if "pytorch3dunet" not in sys.modules:
        p3 = _stub("pytorch3dunet")
        p3u = _stub("pytorch3dunet.unet3d")
        spec = importlib.util.spec_from_file_location("pytorch3dunet.unet3d.se",
                                                      os.path.join(REF, "model/unet3d/se.py"))
        se = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(se)
        sys.modules["pytorch3dunet.unet3d.se"] = se
        p3.unet3d = p3u
        p3u.se = se
        _stub("pytorch3dunet.augment")
    if REF not in sys.path:
        sys.path.insert(0, REF)

    ns = types.SimpleNamespace()
    import model as ref_model  # noqa
    import model.unet2d as unet2d
    import model.unet2d.layers as layers2d
    import model.unet3d.buildingblocks as bb3d
    import model.unet3d.losses as losses3d
    import model.unet3d.model as model3d
    ns.model = ref_model
    ns.unet2d = unet2d
    ns.layers2d = layers2d
    ns.bb3d = bb3d
    ns.losses3d = losses3d
    ns.model3d = model3d
    try:
        import trainer as ref_trainer
        ns.trainer = ref_trainer
    except Exception as e:  # pragma: no cover
        ns.trainer = None
        ns.trainer_error = repr(e)
    try:
        import augment.unet3d_augment.transforms as tr
        sys.modules["pytorch3dunet.augment.transforms"] = tr
        sys.modules["pytorch3dunet.augment"].transforms = tr
        ns.transforms = tr
    except Exception as e:  # pragma: no cover
        ns.transforms = None
        ns.transforms_error = repr(e)
    return ns
