"""Every import statement of the reference's drop-in surface, of its own scripts (train.py:5-9, test_model.py:7,
test_trainer.py:6) and of INTEGRATION.md's examples must run VERBATIM after `dropin.install()`, and must resolve to the package's own
module objects (no second copy under the short name: round 1 re-imported sub-modules and broke their relative imports).
Also: the `TrainingArguments` keyword drift of SURVEY.md §8b (`evaluation_strategy`, `warmup_ratio`, `logging_dir`; reference train.py:120-137)."""
import os
import re
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

# §8b, "Import surface (must keep working unchanged)" - one statement per listed module / name group
SURFACE = """
from unet2d import UNetModel, UNetConfig
from unet2d_dataset import DRIVEDataset, DRIVEDataCollator, BUSIDataset, BUSIDataCollator
from model import UNetModel, UNetConfig
from unet2d import UNet
from trainer import CustomTrainer, compute_metrics
from model.unet2d import UNet, UNetConfig, UNetModel, UNetModelOutput, UNet_3Plus, UNet_3Plus_DeepSup, UNet_3Plus_DeepSup_CGM
from model.unet2d import init_weights, DoubleConvolution, DownSample, UpSample, CropAndConcat, unetConv2, unetUp, unetUp_origin
from model.unet2d.unet import UNet, UNetModel, UNetConfig, UNetModelOutput
from model.unet2d.layers import DoubleConvolution, DownSample, UpSample, CropAndConcat, unetConv2
from model.unet2d.loss import SegmentationLoss
from model.unet3d.model import UNet3D, ResidualUNet3D, ResidualUNetSE3D, UNet2D, ResidualUNet2D, AbstractUNet, get_model
from model.unet3d.buildingblocks import create_conv, SingleConv, DoubleConv, ResNetBlock, Encoder, Decoder
import model.unet3d.losses
from model.unet3d.losses import get_loss_criterion, BCEDiceLoss, DiceLoss
from model.unet3d.UNet3D import UNet3DForMedicalSegmentation, UNet3DForMedicalSegmentationConfig, UNet3DForMedicalSegmentationOutput
from model.unet3d.metrics import MeanIoU, DiceCoefficient
from model.unet3d.predictor import StandardPredictor
from model.unet3d.trainer import UNetTrainer
from augment.unet3d_augment.transforms import Transformer
from augment.unet3d_augment.transforms import RandomFlip, RandomRotate90, RandomRotate, RandomContrast, Standardize, Normalize, AdditiveGaussianNoise
from dataset.unet2d_dataset.MYDataset import DRIVEDataset, BUSIDataset
from dataset.unet2d_dataset.MYDataCollator import DRIVEDataCollator
from dataset.unet3d_dataset.utils import SliceBuilder
from unet3d_dataset.utils import SliceBuilder, FilterSliceBuilder
""".strip().splitlines()


@pytest.fixture(scope="module")
def installed():
    import mdeical_image_segmentation_amd.dropin as d
    d.install()
    return d


@pytest.mark.parametrize("line", SURFACE)
def test_surface_import_line(installed, line):
    exec(line, {})


def _integration_import_lines():
    txt = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    out = []
    for block in re.findall(r"```python\n(.*?)```", txt, flags=re.S):
        for ln in block.splitlines():
            ln = ln.split("#", 1)[0].strip()
            if re.match(r"^(from (model|unet2d|unet3d|trainer|dataset|unet2d_dataset|unet3d_dataset|augment)[\w.]* import |import (model|trainer|augment|dataset)[\w.]*$)", ln):
                out.append(ln)
    return sorted(set(out))


def test_integration_md_import_lines(installed):
    lines = _integration_import_lines()
    assert len(lines) >= 8, lines
    for ln in lines:
        exec(ln, {})


def test_aliases_are_the_same_module_objects(installed):
    import importlib
    for short, real in [("model.unet3d.model", "mdeical_image_segmentation_amd.model.unet3d.model"),
                        ("unet2d.unet", "mdeical_image_segmentation_amd.model.unet2d.unet"),
                        ("augment.unet3d_augment.transforms", "mdeical_image_segmentation_amd.augment.unet3d_augment.transforms"),
                        ("unet2d_dataset.MYDataset", "mdeical_image_segmentation_amd.dataset.unet2d_dataset.MYDataset")]:
        a, b = importlib.import_module(short), importlib.import_module(real)
        assert a is b and a.__name__ == real
        assert sys.modules[short] is b
    with pytest.raises(ModuleNotFoundError):
        importlib.import_module("model.unet3d.no_such_module")


def test_training_arguments_accepts_the_reference_keywords(installed, tmp_path):
    """reference train.py:120-137 verbatim keyword set (transformers 4.40 spelling) on the installed transformers."""
    from transformers import TrainingArguments
    out = tmp_path / "run"
    args = TrainingArguments(
        output_dir=out / "results", evaluation_strategy="steps", eval_steps=100, logging_dir=out / "logs", logging_steps=20,
        num_train_epochs=200, per_device_train_batch_size=1, per_device_eval_batch_size=1, save_steps=100, save_total_limit=2,
        remove_unused_columns=False, label_names=["labels"], warmup_ratio=0.001, learning_rate=0.005, weight_decay=0.001,
        metric_for_best_model="iou", report_to=[])
    assert str(getattr(args, "eval_strategy", getattr(args, "evaluation_strategy", None))).endswith("steps") or \
        getattr(args, "eval_strategy").value == "steps"
    ratio = getattr(args, "warmup_ratio", None)
    if not ratio:
        assert abs(float(args.warmup_steps) - 0.001) < 1e-12      # float in [0, 1) = ratio of the total steps (transformers 5.x)
        assert args.get_warmup_steps(10000) == 10
    assert args.learning_rate == 0.005 and args.weight_decay == 0.001 and args.label_names == ["labels"]
    # the modern spelling still works through the same class
    a2 = TrainingArguments(output_dir=out / "r2", eval_strategy="no", warmup_steps=3, report_to=[])
    assert a2.warmup_steps == 3


def test_compat_training_arguments_pickles(installed, tmp_path):
    """HF Trainer torch.save()s its TrainingArguments into every checkpoint: the compatibility class must be importable by name"""
    import io
    import pickle

    import torch
    from transformers import TrainingArguments
    a = TrainingArguments(output_dir=tmp_path / "o", evaluation_strategy="no", warmup_ratio=0.25, report_to=[])
    b = pickle.loads(pickle.dumps(a))
    assert type(b) is type(a) and float(b.warmup_steps) == 0.25
    torch.save(a, io.BytesIO())
