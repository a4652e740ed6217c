"""Mirror of the helpers of model/unet3d/utils.py that the 3-D path needs: feature widths (:109-110), checkpoint files (:13-58),
`RunningAverage` (:97-106), optimizer / LR-scheduler factories (:275-368), logger.  Host-side control code: plain PyTorch, no kernels."""
import importlib
import logging
import os
import shutil
import sys

import torch
from torch import optim


def number_of_features_per_level(init_channel_number, num_levels):
    return [init_channel_number * 2 ** k for k in range(num_levels)]


def save_checkpoint(state, is_best, checkpoint_dir):
    """'{checkpoint_dir}/last_checkpoint.pytorch' (+ a copy as 'best_checkpoint.pytorch' when is_best) - utils.py:13-32"""
    if not os.path.exists(checkpoint_dir):
        os.mkdir(checkpoint_dir)
    last = os.path.join(checkpoint_dir, "last_checkpoint.pytorch")
    torch.save(state, last)
    if is_best:
        shutil.copyfile(last, os.path.join(checkpoint_dir, "best_checkpoint.pytorch"))


def load_checkpoint(checkpoint_path, model, optimizer=None, model_key="model_state_dict", optimizer_key="optimizer_state_dict"):
    """utils.py:35-58: restores the model (and optimizer) from a checkpoint written by save_checkpoint; returns the stored state dict"""
    if not os.path.exists(checkpoint_path):
        raise IOError(f"Checkpoint '{checkpoint_path}' does not exist")
    state = torch.load(checkpoint_path, map_location="cpu")
    model.load_state_dict(state[model_key])
    if optimizer is not None:
        optimizer.load_state_dict(state[optimizer_key])
    return state


_loggers = {}


def get_logger(name, level=logging.INFO):
    if name not in _loggers:
        logger = logging.getLogger(name)
        logger.setLevel(level)
        handler = logging.StreamHandler(sys.stdout)
        handler.setFormatter(logging.Formatter("%(asctime)s [%(threadName)s] %(levelname)s %(name)s - %(message)s"))
        logger.addHandler(handler)
        _loggers[name] = logger
    return _loggers[name]


def get_number_of_learnable_parameters(model):
    return sum(p.numel() for p in model.parameters() if p.requires_grad)


class RunningAverage:
    """count-weighted running mean (utils.py:97-106)"""

    def __init__(self):
        self.count = 0
        self.sum = 0
        self.avg = 0

    def update(self, value, n=1):
        self.count += n
        self.sum += value * n
        self.avg = self.sum / self.count


# optimizer name -> (class, {constructor keyword: (config key, default)}); 'learning_rate' and 'weight_decay' are common (utils.py:275-352)
_BETAS = {"betas": ("betas", (0.9, 0.999))}
_OPTIMIZERS = {
    "Adadelta": (optim.Adadelta, {"rho": ("rho", 0.9)}),
    "Adagrad": (optim.Adagrad, {"lr_decay": ("lr_decay", 0)}),
    "AdamW": (optim.AdamW, _BETAS),
    "Adamax": (optim.Adamax, _BETAS),
    "NAdam": (optim.NAdam, {**_BETAS, "momentum_decay": ("momentum_decay", 4e-3)}),
    "RAdam": (optim.RAdam, _BETAS),
    "RMSprop": (optim.RMSprop, {"alpha": ("alpha", 0.99)}),
    "Rprop": (optim.RMSprop, {"momentum": ("momentum", 0)}),          # sic: the reference builds RMSprop for 'Rprop' (:337-339)
    "SGD": (optim.SGD, {"momentum": ("momentum", 0), "dampening": ("dampening", 0), "nesterov": ("nesterov", False)}),
    "Adam": (optim.Adam, _BETAS),
}


def create_optimizer(optimizer_config, model):
    name = optimizer_config.get("name", "Adam")
    if name in ("SparseAdam", "ASGD", "LBFGS"):
        raise NotImplementedError(f"create_optimizer: {name} is not wired on the MI355X path")
    cls, extra = _OPTIMIZERS.get(name, _OPTIMIZERS["Adam"])            # unknown names fall back to Adam like the reference
    kw = {k: (tuple(optimizer_config.get(key, dflt)) if k == "betas" else optimizer_config.get(key, dflt)) for k, (key, dflt) in extra.items()}
    return cls(model.parameters(), lr=optimizer_config.get("learning_rate", 1e-3), weight_decay=optimizer_config.get("weight_decay", 0), **kw)


def create_lr_scheduler(lr_config, optimizer):
    """utils.py:355-363 (pops 'name' from the config like the reference)"""
    if lr_config is None:
        return None
    clazz = getattr(importlib.import_module("torch.optim.lr_scheduler"), lr_config.pop("name"))
    lr_config["optimizer"] = optimizer
    return clazz(**lr_config)


class _TensorboardFormatter:
    """utils.py:113-151: batch -> [(tag, CHW image)]"""

    def __init__(self, **kwargs):
        pass

    def __call__(self, name, batch):
        import numpy as np
        out = []
        for tag, img in self.process_batch(name, batch):
            assert img.ndim == 2 or img.ndim == 3, "Only 2D (HW) and 3D (CHW) images are accepted for display"
            if img.ndim == 2:
                img = np.expand_dims(img, axis=0)
            else:
                assert img.shape[0] in (1, 3), "Only (1, H, W) or (3, H, W) images are supported"
            out.append((tag, img))
        return out

    def process_batch(self, name, batch):
        raise NotImplementedError


class DefaultTensorboardFormatter(_TensorboardFormatter):
    """utils.py:154-187: the middle z-slice of every (sample, channel), min-max normalised"""

    def __init__(self, skip_last_target=False, **kwargs):
        super().__init__(**kwargs)
        self.skip_last_target = skip_last_target

    def process_batch(self, name, batch):
        import numpy as np
        if name == "targets" and self.skip_last_target:
            batch = batch[:, :-1, ...]
        norm = lambda img: np.nan_to_num((img - np.min(img)) / np.ptp(img))
        out = []
        if batch.ndim == 5:
            z = batch.shape[2] // 2
            for b in range(batch.shape[0]):
                for c in range(batch.shape[1]):
                    out.append((f"{name}/batch_{b}/channel_{c}/slice_{z}", norm(batch[b, c, z, ...])))
        else:
            z = batch.shape[1] // 2
            for b in range(batch.shape[0]):
                out.append((f"{name}/batch_{b}/channel_0/slice_{z}", norm(batch[b, z, ...])))
        return out


def get_tensorboard_formatter(formatter_config):
    """utils.py:212-219"""
    if formatter_config is None:
        return DefaultTensorboardFormatter()
    cfg = dict(formatter_config)
    clazz = globals()[cfg["name"]]
    return clazz(**cfg)
