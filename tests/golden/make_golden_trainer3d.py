"""G12: the reference's native 3-D loop (model/unet3d/trainer.py `UNetTrainer`) on a tiny real UNet3D, CPU, build container only.

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_trainer3d.py

Absent third-party packages the import chain touches (tensorboard, albumentations, imageio, debugpy, skimage.metrics, pytorch3dunet = the
upstream of the local model/unet3d files) get in-process stand-ins; the SummaryWriter stand-in records the scalars the trainer logs.
Stores the batches, every logged scalar, the final counters, the checkpoint's keys / counters and statistics of the trained parameters."""
import os
import sys
import tempfile

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from _ref_import import _stub  # noqa: E402
from make_golden_metrics3d import load_ref_metrics  # noqa: E402

torch.set_num_threads(8)


class _Writer:
    last = None

    def __init__(self, *a, **k):
        self.scalars = []
        _Writer.last = self

    def add_scalar(self, tag, value, it):
        self.scalars.append((tag, float(value), int(it)))

    def add_image(self, *a, **k):
        pass

    def add_histogram(self, *a, **k):
        pass


def load_ref_trainer():
    M = load_ref_metrics()
    tb = _stub("torch.utils.tensorboard", SummaryWriter=_Writer)
    import torch.utils
    torch.utils.tensorboard = tb
    sys.modules["model.unet3d.metrics"] = M
    for name in ("albumentations", "albumentations.pytorch", "imageio", "debugpy"):
        if name not in sys.modules:
            _stub(name)
    sys.modules["albumentations.pytorch"].ToTensorV2 = None
    import model.unet3d.trainer as T
    return T, M


def main():
    T, M = load_ref_trainer()
    import model.unet3d.losses as L
    import model.unet3d.model as MM
    import model.unet3d.utils as U
    torch.manual_seed(0)
    net = MM.UNet3D(1, 3, f_maps=[64, 128], num_levels=2)
    rng = np.random.RandomState(12)
    tr = [(rng.randn(1, 1, 16, 16, 16).astype(np.float32), (rng.rand(1, 3, 16, 16, 16) > 0.5).astype(np.float32)) for _ in range(4)]
    va = [(rng.randn(1, 1, 16, 16, 16).astype(np.float32), (rng.rand(1, 3, 16, 16, 16) > 0.5).astype(np.float32)) for _ in range(2)]
    loaders = {"train": [(torch.from_numpy(a), torch.from_numpy(b)) for a, b in tr], "val": [(torch.from_numpy(a), torch.from_numpy(b)) for a, b in va]}
    opt = U.create_optimizer({"learning_rate": 1e-3, "weight_decay": 1e-5}, net)
    sched = U.create_lr_scheduler({"name": "StepLR", "step_size": 1, "gamma": 0.5}, opt)
    loss = L.get_loss_criterion({"loss": {"name": "BCEDiceLoss"}})
    ev = M.get_evaluation_metric({"eval_metric": {"name": "MeanIoU"}})
    with tempfile.TemporaryDirectory() as d:
        t = T.UNetTrainer(net, opt, sched, loss, ev, loaders, checkpoint_dir=d, max_num_epochs=3, max_num_iterations=6, validate_after_iters=2,
                          log_after_iters=1, tensorboard_formatter=lambda name, batch: [])
        t.fit()
        files = sorted(f for f in os.listdir(d) if f.endswith(".pytorch"))
        last = torch.load(os.path.join(d, "last_checkpoint.pytorch"), map_location="cpu")
        best = torch.load(os.path.join(d, "best_checkpoint.pytorch"), map_location="cpu")
    sc = _Writer.last.scalars
    out = {"train_x": np.stack([a for a, _ in tr]), "train_t": np.stack([b for _, b in tr]), "val_x": np.stack([a for a, _ in va]),
           "val_t": np.stack([b for _, b in va]),
           "scalar_tags": np.array([s[0] for s in sc]), "scalar_values": np.array([s[1] for s in sc]), "scalar_iters": np.array([s[2] for s in sc]),
           "num_iterations": np.array(t.num_iterations), "num_epochs": np.array(t.num_epochs), "best_eval_score": np.array(t.best_eval_score),
           "files": np.array(files), "ckpt_keys": np.array(sorted(last.keys())),
           "last_counters": np.array([last["num_epochs"], last["num_iterations"]]), "last_best": np.array(last["best_eval_score"]),
           "best_counters": np.array([best["num_epochs"], best["num_iterations"]]),
           "state_keys": np.array(list(last["model_state_dict"].keys())),
           "param_stats": np.array([[v.double().sum().item(), v.double().abs().sum().item()] for v in last["model_state_dict"].values()]),
           "final_lr": np.array(opt.param_groups[0]["lr"])}
    np.savez_compressed(os.path.join(HERE, "g12_trainer3d.npz"), **out)
    for s in sc:
        print(s)
    print("iters", t.num_iterations, "epochs", t.num_epochs, "best", t.best_eval_score, files, out["last_counters"], out["best_counters"])


if __name__ == "__main__":
    main()
