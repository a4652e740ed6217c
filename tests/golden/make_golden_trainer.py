"""G7: a tiny end-to-end run of the REAL reference through HF Trainer (build container, CPU): 4 steps, bs 2, 32x32.

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_trainer.py

transformers 5.x calls compute_loss(..., num_items_in_batch=...); the reference's 4.40-era signature lacks it, so the
reference CustomTrainer is subclassed with exactly that extra keyword (SURVEY.md §8b "version drift").
"""
import os
import sys
import tempfile

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from _ref_import import import_reference  # noqa: E402


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
    """dataset/unet2d_dataset/MYDataCollator.py:7-15: stack into {"images", "labels"}"""
    return {"images": torch.stack([b["images"] for b in batch]), "labels": torch.stack([b["labels"] for b in batch])}


def run(UNetModel, UNetConfig, CustomTrainer, device_args, out_dir):
    from transformers import TrainingArguments
    torch.manual_seed(0)
    model = UNetModel(UNetConfig(in_channels=1, out_channels=1, unet_type="UNet"))
    args = TrainingArguments(output_dir=out_dir, per_device_train_batch_size=2, max_steps=4, learning_rate=5e-3, weight_decay=1e-3,
                             logging_steps=1, save_strategy="no", report_to=[], remove_unused_columns=False, label_names=["labels"],
                             seed=42, dataloader_num_workers=0, max_grad_norm=1.0, lr_scheduler_type="linear", warmup_steps=0,
                             **device_args)
    tr = CustomTrainer(model=model, args=args, train_dataset=SynthDataset(), data_collator=collate)
    tr.train()
    losses = [h["loss"] for h in tr.state.log_history if "loss" in h]
    gnorm = [h.get("grad_norm", float("nan")) for h in tr.state.log_history if "loss" in h]
    return model, losses, gnorm


def main():
    ns = import_reference()
    Base = ns.trainer.CustomTrainer

    class Shim(Base):
        def compute_loss(self, model, inputs, return_outputs=False, num_items_in_batch=None):
            return super().compute_loss(model, inputs, return_outputs)

    with tempfile.TemporaryDirectory() as d:
        model, losses, gnorm = run(ns.unet2d.UNetModel, ns.unet2d.UNetConfig, Shim, {"use_cpu": True}, d)
    fin = model.unet.final_conv.weight.detach().flatten().numpy()
    np.savez_compressed(os.path.join(HERE, "g7_trainer.npz"), losses=np.array(losses), grad_norm=np.array(gnorm), final_conv_w=fin)
    print("losses", losses, "grad_norm", gnorm)


if __name__ == "__main__":
    main()
