"""Make the reference's import surface resolve to this package:

    import mdeical_image_segmentation_amd.dropin as d; d.install()
    from unet2d import UNetModel, UNetConfig                       # train.py:5   (model/ on sys.path)
    from unet2d_dataset import DRIVEDataset, DRIVEDataCollator     # train.py:6   (dataset/ on sys.path)
    from trainer import CustomTrainer, compute_metrics             # train.py:9
    from model import UNetModel                                    # test_trainer.py:6
    from model.unet3d.model import UNet3D                          # model/unet3d/model.py
    from model.unet3d.losses import get_loss_criterion             # model/unet3d/losses.py:273
    from model.unet3d.UNet3D import UNet3DForMedicalSegmentation   # model/unet3d/UNet3D.py
    from augment.unet3d_augment.transforms import Transformer      # augment/unet3d_augment/transforms.py:721

Every reference-style dotted name below one of the alias roots is mapped onto THE SAME module object as its
`mdeical_image_segmentation_amd.*` original by a `sys.meta_path` finder, so sub-modules keep their real `__name__` /
`__package__` and their relative imports keep working (aliasing only the package objects, as round 1 did, re-imported
sub-modules under the short name and broke `from ..._lib import`).

`install()` also closes the transformers version drift of SURVEY.md §8b: the reference's `train.py:120-137` passes
`evaluation_strategy=`, `warmup_ratio=` and `logging_dir=` to `TrainingArguments`; transformers >= 4.46 / 5.x renamed or
dropped them.  `TrainingArguments` here accepts both spellings, and `install(patch_transformers=True)` (the default) puts it
in place of `transformers.TrainingArguments` when the installed class lacks the old names, so the user's script runs unchanged.
"""
import importlib
import importlib.abc
import importlib.machinery
import sys

_PKG = __name__.rsplit(".", 1)[0]

# alias root -> real package (the reference puts model/ and dataset/ on sys.path, hence the bare `unet2d`, `unet2d_dataset`)
ALIASES = {
    "model": f"{_PKG}.model",
    "unet2d": f"{_PKG}.model.unet2d",
    "unet3d": f"{_PKG}.model.unet3d",
    "trainer": f"{_PKG}.trainer",
    "dataset": f"{_PKG}.dataset",
    "unet2d_dataset": f"{_PKG}.dataset.unet2d_dataset",
    "unet3d_dataset": f"{_PKG}.dataset.unet3d_dataset",
    "augment": f"{_PKG}.augment",
    "unet3d_augment": f"{_PKG}.augment.unet3d_augment",
}


def _real_name(fullname):
    root, _, rest = fullname.partition(".")
    target = ALIASES.get(root)
    if target is None:
        return None
    return target + ("." + rest if rest else "")


class _AliasLoader(importlib.abc.Loader):
    def __init__(self, real):
        self.real = real

    def create_module(self, spec):
        return importlib.import_module(self.real)     # the SAME module object, not a copy

    def exec_module(self, module):
        pass


class _AliasFinder(importlib.abc.MetaPathFinder):
    def find_spec(self, fullname, path=None, target=None):
        real = _real_name(fullname)
        if real is None:
            return None
        try:
            mod = importlib.import_module(real)
        except ModuleNotFoundError as e:
            if e.name is not None and (e.name == real or real.startswith(e.name + ".")):
                return None                          # no such mirror module: let the normal machinery report it
            raise
        return importlib.machinery.ModuleSpec(fullname, _AliasLoader(real), is_package=hasattr(mod, "__path__"))


_finder = None


def _compat_kwargs(kw, fields):
    """Map the reference's `TrainingArguments` keywords (transformers 4.40, requirements.txt:176) onto the installed version's."""
    kw = dict(kw)
    if "evaluation_strategy" in kw and "evaluation_strategy" not in fields:
        v = kw.pop("evaluation_strategy")
        kw.setdefault("eval_strategy", v)
    if "warmup_ratio" in kw and "warmup_ratio" not in fields:
        r = kw.pop("warmup_ratio")
        # transformers 5.x: a float warmup_steps in [0, 1) IS the ratio of the total steps
        if r and not kw.get("warmup_steps"):
            kw["warmup_steps"] = float(r)
    if "logging_dir" in kw and "logging_dir" not in fields:
        import os
        os.environ.setdefault("TENSORBOARD_LOGGING_DIR", str(kw.pop("logging_dir")))
    return kw


def _make_training_arguments():
    import dataclasses

    import transformers
    if "TrainingArguments" in globals():
        return globals()["TrainingArguments"]
    base = transformers.TrainingArguments
    if getattr(base, "_misamd_compat", False):
        return base
    fields = {f.name for f in dataclasses.fields(base)}
    if {"evaluation_strategy", "warmup_ratio", "logging_dir"} <= fields:
        return base                                   # the reference's own pin: nothing to translate

    class TrainingArguments(base):
        _misamd_compat = True

        def __init__(self, *args, **kw):
            super().__init__(*args, **_compat_kwargs(kw, fields))

    # importable (hence picklable: HF Trainer torch.save()s its args into every checkpoint) as <this module>.TrainingArguments
    TrainingArguments.__qualname__ = TrainingArguments.__name__ = "TrainingArguments"
    TrainingArguments.__module__ = __name__
    globals()["TrainingArguments"] = TrainingArguments
    return TrainingArguments


def __getattr__(name):
    if name == "TrainingArguments":
        return _make_training_arguments()
    raise AttributeError(name)


def install(patch_transformers=True):
    global _finder
    if _finder is None:
        _finder = _AliasFinder()
        sys.meta_path.insert(0, _finder)
    for alias in ALIASES:                              # eager for the roots (an already-imported unrelated `model` must not win)
        sys.modules[alias] = importlib.import_module(ALIASES[alias])
    if patch_transformers:
        import transformers
        ta = _make_training_arguments()
        if ta is not transformers.TrainingArguments:
            transformers.TrainingArguments = ta


def uninstall():
    global _finder
    if _finder is not None:
        sys.meta_path.remove(_finder)
        _finder = None
    for name in [n for n in sys.modules if n.split(".", 1)[0] in ALIASES]:
        del sys.modules[name]
