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
PIPELINE = len(sys.argv) > 2 and sys.argv[2] == "pipeline"       # uint8 DRIVE-sized samples in HBM + the device sample pipeline (dataset mirror), 3 -> 1 channels, BCE
DEVICE_DATA = len(sys.argv) > 2 and sys.argv[2] in ("device", "pipeline")      # keep the (decoded) dataset in HBM: items are CUDA tensors, no workers, no H2D
B, S = 32, 512


class Synth(torch.utils.data.Dataset):
    def __init__(self, n):
        g = torch.Generator().manual_seed(0)
        self.images = torch.randn(n, 1, S, S, generator=g)
        self.labels = torch.randint(0, 2, (n, S, S), generator=g)
        if DEVICE_DATA:
            self.images, self.labels = self.images.cuda(), self.labels.cuda()

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
if PIPELINE:
    import numpy as np
    from unet2d_dataset import DeviceSegmentationDataset, DRIVEDataCollator
    r = np.random.RandomState(0)
    train_ds = DeviceSegmentationDataset([r.randint(0, 256, (584, 565, 3)).astype(np.uint8) for _ in range(4 * B)],
                                         [(r.rand(584, 565) > 0.8).astype(np.uint8) * 255 for _ in range(4 * B)], train=True)
    collate = DRIVEDataCollator()
    model = UNetModel(UNetConfig(in_channels=3, out_channels=1, unet_type="UNet"))
else:
    train_ds = None
    model = UNetModel(UNetConfig(in_channels=1, out_channels=2, unet_type="UNet"))
clock = Clock()
with tempfile.TemporaryDirectory() as out:
    args = TrainingArguments(output_dir=out, per_device_train_batch_size=B, max_steps=steps, learning_rate=5e-3, weight_decay=1e-3, logging_steps=10,
                             save_strategy="no", report_to=[], remove_unused_columns=False, label_names=["labels"], seed=42,
                             dataloader_num_workers=0 if DEVICE_DATA else 4, dataloader_pin_memory=not DEVICE_DATA, max_grad_norm=1.0)
    tr = CustomTrainer(model=model, args=args, train_dataset=train_ds if PIPELINE else Synth(4 * B), data_collator=collate, callbacks=[clock])
    tr.train()
dt = (clock.t[-1] - clock.t[9]) / (len(clock.t) - 10)
print(("uint8 samples in HBM + device sample pipeline: " if PIPELINE else "device-resident dataset: " if DEVICE_DATA else "host dataset: ") + f"CustomTrainer {os.environ.get('MISAMD_DTYPE', 'f32')} bs={B} {S}x{S}: {dt * 1e3:.1f} ms/step = {B / dt:.1f} images/s (steps 11..{len(clock.t)})")
