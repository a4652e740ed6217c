"""End-to-end drop-in: the mirror `CustomTrainer` (HF Trainer) + mirror `UNetModel` on MI355X against a run of the REAL
reference through the same HF Trainer on CPU (tests/golden/g7_trainer.npz, made by tests/golden/make_golden_trainer.py)."""
import tempfile

import numpy as np
import pytest
import torch

from conftest import load_golden


class SynthDataset(torch.utils.data.Dataset):
    def __init__(self, n=8, size=32, seed=5):
        g = torch.Generator().manual_seed(seed)
        self.images = torch.rand(n, 1, size, size, generator=g)
        self.labels = (torch.rand(n, 1, size, size, generator=g) > 0.5).float()

    def __len__(self):
        return len(self.images)

    def __getitem__(self, i):
        return {"images": self.images[i], "labels": self.labels[i]}


def collate(batch):
    return {"images": torch.stack([b["images"] for b in batch]), "labels": torch.stack([b["labels"] for b in batch])}


@pytest.mark.gpu
def test_custom_trainer_reproduces_reference_run():
    from transformers import TrainingArguments

    import mdeical_image_segmentation_amd.dropin as d
    d.install()
    from trainer import CustomTrainer
    from unet2d import UNetConfig, UNetModel
    g = load_golden("g7_trainer.npz")
    torch.manual_seed(0)
    model = UNetModel(UNetConfig(in_channels=1, out_channels=1, unet_type="UNet"))
    with tempfile.TemporaryDirectory() as out:
        args = TrainingArguments(output_dir=out, per_device_train_batch_size=2, max_steps=4, learning_rate=5e-3, weight_decay=1e-3,
                                 logging_steps=1, save_strategy="no", report_to=[], remove_unused_columns=False, label_names=["labels"],
                                 seed=42, dataloader_num_workers=0, max_grad_norm=1.0, lr_scheduler_type="linear", warmup_steps=0)
        from trainer import compute_metrics
        tr = CustomTrainer(model=model, args=args, train_dataset=SynthDataset(), eval_dataset=SynthDataset(n=4, seed=6), data_collator=collate,
                           compute_metrics=compute_metrics)
        tr.train()
        # trainer.evaluate() (train.py:160): eval-mode forward through the engine, logits gathered by HF, metrics by the device kernel
        ev = tr.evaluate()
        assert "eval_iou" in ev and "eval_dice" in ev and 0.0 <= ev["eval_iou"] <= 1.0 and 0.0 <= ev["eval_dice"] <= 1.0
        from oracle import metrics_oracle as mo
        eds = SynthDataset(n=4, seed=6)
        with torch.no_grad():
            lg = model(images=eds.images.cuda(), labels=eds.labels.cuda()).logits.cpu().numpy()
        ref = mo.compute_metrics(lg, eds.labels.numpy())
        assert abs(ev["eval_iou"] - float(ref["iou"])) < 2e-3 and abs(ev["eval_dice"] - float(ref["dice"])) < 2e-3, (ev, ref)
        # checkpoint round trip with the reference's key layout (save_pretrained -> from_pretrained)
        model.save_pretrained(out + "/ckpt")
        m2 = UNetModel.from_pretrained(out + "/ckpt").cuda().eval()
        with torch.no_grad():
            lg2 = m2(images=eds.images.cuda(), labels=eds.labels.cuda()).logits.cpu().numpy()
        assert np.array_equal(lg, lg2), "logits differ after the checkpoint round trip"
    losses = [h["loss"] for h in tr.state.log_history if "loss" in h]
    gnorm = [h["grad_norm"] for h in tr.state.log_history if "loss" in h]
    ref_l, ref_g = g["losses"], g["grad_norm"]
    print("losses", losses, "reference", ref_l.tolist())
    assert len(losses) == 4
    # steps 1-2: before the (chaotic, lr 5e-3) trajectory can amplify fp32 differences
    assert abs(losses[0] - ref_l[0]) < 1e-4 and abs(losses[1] - ref_l[1]) < 1e-3
    assert abs(gnorm[0] - ref_g[0]) < 1e-3 * max(1.0, ref_g[0])
    assert abs(losses[2] - ref_l[2]) < 2e-2 and abs(losses[3] - ref_l[3]) < 0.15 * ref_l[3]
    # the checkpoint surface: the state dict round-trips through safetensors-style keys
    sd = model.state_dict()
    assert len(sd) == 46 and all(k.startswith("unet.") for k in sd)


@pytest.mark.gpu
def test_reference_train_py_block_runs_unchanged(tmp_path):
    """The body of the reference's train.py:110-160 with its own argparse defaults (transformers 4.40 keyword spelling:
    evaluation_strategy / warmup_ratio / logging_dir), a synthetic dataset instead of DRIVE, 2 optimizer steps + evaluate().
    `dropin.install()` is the only line a user adds."""
    import argparse

    import mdeical_image_segmentation_amd.dropin as d
    d.install()
    from pathlib import Path

    from trainer import CustomTrainer, compute_metrics
    from transformers import TrainingArguments
    from unet2d import UNetConfig, UNetModel

    args = argparse.Namespace(output_dir=str(tmp_path), evaluation_strategy="steps", eval_steps=100, logging_steps=1, num_train_epochs=5000,
                              per_device_train_batch_size=1, per_device_eval_batch_size=1, save_steps=1000, save_total_limit=5,
                              warmup_ratio=0.001, learning_rate=0.005, weight_decay=0.001, metric_for_best_model="iou", in_channels=1,
                              out_channels=1, unet_type="UNet")
    output_dir = Path(args.output_dir).joinpath("run").joinpath(args.unet_type)
    output_dir.mkdir(exist_ok=True, parents=True)
    training_args = TrainingArguments(
        output_dir=output_dir.joinpath("results"),
        evaluation_strategy=args.evaluation_strategy,
        eval_steps=args.eval_steps,
        logging_dir=output_dir.joinpath("logs"),
        logging_steps=args.logging_steps,
        num_train_epochs=args.num_train_epochs,
        per_device_train_batch_size=args.per_device_train_batch_size,
        per_device_eval_batch_size=args.per_device_eval_batch_size,
        save_steps=args.save_steps,
        save_total_limit=args.save_total_limit,
        remove_unused_columns=False,
        label_names=["labels"],
        warmup_ratio=args.warmup_ratio,
        learning_rate=args.learning_rate,
        weight_decay=args.weight_decay,
        metric_for_best_model=args.metric_for_best_model,
        max_steps=2, report_to=[],            # the only additions: stop after two steps, no tensorboard in the test
    )
    config = UNetConfig(in_channels=args.in_channels, out_channels=args.out_channels, unet_type=args.unet_type)
    model = UNetModel(config)
    trainer = CustomTrainer(model=model, args=training_args, train_dataset=SynthDataset(n=4), eval_dataset=SynthDataset(n=2, seed=6),
                            data_collator=collate, compute_metrics=compute_metrics)
    trainer.train()
    ev = trainer.evaluate()
    losses = [h["loss"] for h in trainer.state.log_history if "loss" in h]
    assert len(losses) == 2 and all(np.isfinite(losses))
    assert "eval_iou" in ev and "eval_dice" in ev
    assert trainer.state.global_step == 2
