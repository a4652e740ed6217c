"""The native 3-D loop end to end on the HIP path: `model/unet3d/trainer.py::UNetTrainer` mirror (UNet3D mirror + BCEDiceLoss + MeanIoU kernels,
torch Adam + StepLR, checkpoints) against the scalars, counters and checkpoint the REAL reference trainer produced on the same batches
(tests/golden/g12_trainer3d.npz), then a resume from the written checkpoint."""
import os

import numpy as np
import pytest
import torch

from conftest import load_golden

pytestmark = pytest.mark.gpu


def _parts(g):
    from mdeical_image_segmentation_amd.model.unet3d import utils as U
    from mdeical_image_segmentation_amd.model.unet3d.losses import get_loss_criterion
    from mdeical_image_segmentation_amd.model.unet3d.metrics import get_evaluation_metric
    from mdeical_image_segmentation_amd.model.unet3d.model import get_model
    torch.manual_seed(0)
    net = get_model({"name": "UNet3D", "in_channels": 1, "out_channels": 3, "f_maps": [64, 128], "num_levels": 2}).cuda()
    opt = U.create_optimizer({"learning_rate": 1e-3, "weight_decay": 1e-5}, net)
    sched = U.create_lr_scheduler({"name": "StepLR", "step_size": 1, "gamma": 0.5}, opt)
    loss = get_loss_criterion({"loss": {"name": "BCEDiceLoss"}})
    ev = get_evaluation_metric({"eval_metric": {"name": "MeanIoU"}})
    T = torch.from_numpy
    loaders = {"train": [(T(a), T(b)) for a, b in zip(g["train_x"], g["train_t"])], "val": [(T(a), T(b)) for a, b in zip(g["val_x"], g["val_t"])]}
    return net, opt, sched, loss, ev, loaders


def test_unet_trainer_matches_reference_run(tmp_path):
    from mdeical_image_segmentation_amd.model.unet3d.trainer import UNetTrainer
    g = load_golden("g12_trainer3d.npz")
    net, opt, sched, loss, ev, loaders = _parts(g)
    t = UNetTrainer(net, opt, sched, loss, ev, loaders, checkpoint_dir=str(tmp_path), max_num_epochs=3, max_num_iterations=6,
                    validate_after_iters=2, log_after_iters=1)
    t.fit()
    tags = [s[0] for s in t.scalars]
    assert tags == list(g["scalar_tags"]) and [s[2] for s in t.scalars] == list(g["scalar_iters"])
    for (tag, v, it), ref in zip(t.scalars, g["scalar_values"]):
        if tag == "learning_rate":
            assert v == ref
        elif tag.endswith("loss_avg"):
            assert abs(v - ref) < 2e-3 * abs(ref), (tag, it, v, ref)
        else:       # mean IoU of an untrained net: a channel arg-max over near-tied logits, sensitive to 1e-6 differences
            assert abs(v - ref) < 3e-2, (tag, it, v, ref)
    assert (t.num_iterations, t.num_epochs) == (int(g["num_iterations"]), int(g["num_epochs"]))
    assert abs(t.best_eval_score - float(g["best_eval_score"])) < 3e-2
    assert sorted(f for f in os.listdir(tmp_path) if f.endswith(".pytorch")) == list(g["files"])
    last = torch.load(tmp_path / "last_checkpoint.pytorch", map_location="cpu")
    assert sorted(last.keys()) == list(g["ckpt_keys"])
    assert [last["num_epochs"], last["num_iterations"]] == list(g["last_counters"])
    assert list(last["model_state_dict"].keys()) == list(g["state_keys"])
    stats = np.array([[v.double().sum().item(), v.double().abs().sum().item()] for v in last["model_state_dict"].values()])
    # Adam moves every element by ~lr per step whatever the gradient's magnitude, so elements whose gradient is rounding noise (GroupNorm biases
    # that start at 0, ...) legitimately differ by a few lr: tolerance = 2e-3 relative + 10 % of the summed step sizes (3.5e-3) per element
    numel = np.array([v.numel() for v in last["model_state_dict"].values()], dtype=np.float64)
    tol = 2e-3 * g["param_stats"][:, 1] + 0.1 * 3.5e-3 * numel
    assert (np.abs(stats[:, 1] - g["param_stats"][:, 1]) < tol).all(), "trained parameters drifted from the reference run"
    assert opt.param_groups[0]["lr"] == float(g["final_lr"])

    # resume: counters / best score / weights come back from the checkpoint
    net2, opt2, sched2, loss2, ev2, loaders2 = _parts(g)
    t2 = UNetTrainer(net2, opt2, sched2, loss2, ev2, loaders2, checkpoint_dir=str(tmp_path / "other"), max_num_epochs=3, max_num_iterations=6,
                     validate_after_iters=2, log_after_iters=1, resume=str(tmp_path / "last_checkpoint.pytorch"))
    assert (t2.num_iterations, t2.num_epochs, t2.best_eval_score) == (last["num_iterations"], last["num_epochs"], last["best_eval_score"])
    assert t2.checkpoint_dir == str(tmp_path)
    for (k, a), b in zip(net2.state_dict().items(), last["model_state_dict"].values()):
        assert torch.equal(a.cpu(), b), k


def test_create_trainer_from_config(tmp_path):
    """`create_trainer(config)` (reference model/unet3d/trainer.py:19-55): model, loss, metric, file-backed loaders (dataset/unet3d_dataset/hdf5.py
    `create_datasets`, utils.py:182-227 `get_train_loaders`), optimizer and scheduler from one dictionary; then two epochs of the loop on the HIP path."""
    from mdeical_image_segmentation_amd.model.unet3d.trainer import UNetTrainer, create_trainer
    rng = np.random.RandomState(0)
    paths = {}
    for name in ("train", "val"):
        raw = rng.randn(16, 32, 64).astype(np.float32)
        label = (raw + 0.3 * rng.randn(16, 32, 64) > 0.2).astype(np.float32)
        paths[name] = str(tmp_path / f"{name}.npz")
        np.savez(paths[name], raw=raw, label=label)

    def phase(p, aug):
        raw_t = [{"name": "Standardize"}] + ([{"name": "RandomFlip"}] if aug else []) + [{"name": "ToTensor", "expand_dims": True}]
        lab_t = ([{"name": "RandomFlip"}] if aug else []) + [{"name": "ToTensor", "expand_dims": True}]
        return {"file_paths": [p], "slice_builder": {"name": "SliceBuilder", "patch_shape": [16, 32, 32], "stride_shape": [16, 32, 32], "skip_shape_check": True},
                "transformer": {"raw": raw_t, "label": lab_t}}

    config = {
        "device": "cuda",
        "model": {"name": "UNet3D", "in_channels": 1, "out_channels": 1, "f_maps": [64, 128], "num_levels": 2, "layer_order": "gcr", "final_sigmoid": True},
        "loss": {"name": "BCEDiceLoss"},
        "eval_metric": {"name": "DiceCoefficient"},
        "optimizer": {"learning_rate": 2e-3, "weight_decay": 1e-5},
        "lr_scheduler": {"name": "StepLR", "step_size": 2, "gamma": 0.5},
        "loaders": {"dataset": "StandardHDF5Dataset", "batch_size": 1, "num_workers": 2, "raw_internal_path": "raw", "label_internal_path": "label",
                    "global_normalization": True, "train": phase(paths["train"], True), "val": phase(paths["val"], False)},
        "trainer": {"checkpoint_dir": str(tmp_path / "ckpt"), "max_num_epochs": 2, "max_num_iterations": 100, "validate_after_iters": 2, "log_after_iters": 1,
                    "eval_score_higher_is_better": True},
    }
    t = create_trainer(config)
    assert isinstance(t, UNetTrainer) and len(t.loaders["train"]) == 2 and len(t.loaders["val"]) == 2
    x, y = next(iter(t.loaders["val"]))
    assert x.is_cuda and tuple(x.shape) == (1, 1, 16, 32, 32) and tuple(y.shape) == (1, 1, 16, 32, 32)
    t.fit()
    assert t.num_epochs == 2 and t.num_iterations == 5          # the counter starts at 1 (trainer.py:108) and advances once per batch
    losses = [v for tag, v, _ in t.scalars if tag == "train_loss_avg"]
    assert len(losses) == 4 and all(np.isfinite(losses)) and len(set(losses)) == 4      # (the loop's numerics are pinned by the golden run above)
    assert any(tag == "val_eval_score_avg" for tag, _, _ in t.scalars)
    assert os.path.exists(tmp_path / "ckpt" / "last_checkpoint.pytorch") and os.path.exists(tmp_path / "ckpt" / "best_checkpoint.pytorch")
    with pytest.raises(ImportError, match="h5py"):
        from mdeical_image_segmentation_amd.dataset.unet3d_dataset.hdf5 import read_volumes
        (tmp_path / "x.h5").write_bytes(b"")
        read_volumes(str(tmp_path / "x.h5"), ["raw"])
