"""What a user of the reference sees: the mirror CustomTrainer (HF Trainer: collate, H2D copy, fused engine step through autograd, clip, torch AdamW,
scheduler, logging) at the headline shape, next to the bare engine loop of bench.py.   MISAMD_DTYPE=bf16 python scripts/bench_trainer.py [steps]"""
import os
import sys
import tempfile
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mdeical_image_segmentation_amd.dropin as d  # noqa: E402

d.install()
from trainer import CustomTrainer  # noqa: E402
from transformers import TrainerCallback, TrainingArguments  # noqa: E402
from unet2d import UNetConfig, UNetModel  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 30
B, S = 32, 512


class Synth(torch.utils.data.Dataset):
    def __init__(self, n):
        g = torch.Generator().manual_seed(0)
        self.images = torch.randn(n, 1, S, S, generator=g)
        self.labels = torch.randint(0, 2, (n, S, S), generator=g)

    def __len__(self):
        return len(self.images)

    def __getitem__(self, i):
        return {"images": self.images[i], "labels": self.labels[i]}


def collate(batch):
    return {"images": torch.stack([b["images"] for b in batch]), "labels": torch.stack([b["labels"] for b in batch])}


class Clock(TrainerCallback):
    def __init__(self):
        self.t = []

    def on_step_end(self, args, state, control, **kw):
        torch.cuda.synchronize()
        self.t.append(time.perf_counter())


torch.manual_seed(0)
model = UNetModel(UNetConfig(in_channels=1, out_channels=2, unet_type="UNet"))
clock = Clock()
with tempfile.TemporaryDirectory() as out:
    args = TrainingArguments(output_dir=out, per_device_train_batch_size=B, max_steps=steps, learning_rate=5e-3, weight_decay=1e-3, logging_steps=10,
                             save_strategy="no", report_to=[], remove_unused_columns=False, label_names=["labels"], seed=42,
                             dataloader_num_workers=4, dataloader_pin_memory=True, max_grad_norm=1.0)
    tr = CustomTrainer(model=model, args=args, train_dataset=Synth(4 * B), data_collator=collate, callbacks=[clock])
    tr.train()
dt = (clock.t[-1] - clock.t[9]) / (len(clock.t) - 10)
print(f"CustomTrainer {os.environ.get('MISAMD_DTYPE', 'f32')} bs={B} {S}x{S}: {dt * 1e3:.1f} ms/step = {B / dt:.1f} images/s (steps 11..{len(clock.t)})")
