"""Mirror of the reference's native 3-D training loop (model/unet3d/trainer.py:58-404 `UNetTrainer`): same constructor, same iteration /
validation / LR-scheduler / best-score / checkpoint semantics (`last_checkpoint.pytorch`, `best_checkpoint.pytorch` with the keys num_epochs,
num_iterations, model_state_dict, best_eval_score, optimizer_state_dict) around the MI355X model, loss and metric mirrors.  Control code only:
every tensor op it triggers runs in the HIP engine (model forward/backward), the loss kernels and the metric kernels.  TensorBoard is optional
here (the package is absent in this image): scalars are always kept in `self.scalars` and forwarded to a SummaryWriter when one can be built.
`create_trainer(config)` (trainer.py:19-55) assembles model, loss, metric, loaders, optimizer and scheduler from one configuration dictionary; where the reference
wraps the model in nn.DataParallel (:23-24) the MI355X design is one process per GPU with the gradient all-reduce of ddp.py."""
import os
from datetime import datetime

import torch
from torch import nn
from torch.optim.lr_scheduler import ReduceLROnPlateau

from . import utils
from .utils import get_logger

logger = get_logger("UNetTrainer")


def create_trainer(config):
    """trainer.py:19-55"""
    from ...dataset.unet3d_dataset.utils import get_train_loaders
    from .losses import get_loss_criterion
    from .metrics import get_evaluation_metric
    from .model import get_model
    model = get_model(config["model"])
    if str(config.get("device", "cuda")) == "cpu":
        raise NotImplementedError("create_trainer: device 'cpu' - the MI355X mirror has no CPU path")
    model = model.cuda()
    logger.info(f"Number of learnable params {utils.get_number_of_learnable_parameters(model)}")
    loss_criterion = get_loss_criterion(config)
    eval_criterion = get_evaluation_metric(config)
    loaders = get_train_loaders(config)
    optimizer = utils.create_optimizer(config["optimizer"], model)
    lr_scheduler = utils.create_lr_scheduler(config.get("lr_scheduler", None), optimizer)
    trainer_config = config["trainer"]
    tensorboard_formatter = utils.get_tensorboard_formatter(trainer_config.pop("tensorboard_formatter", None))
    resume = trainer_config.pop("resume", None)
    pre_trained = trainer_config.pop("pre_trained", None)
    return UNetTrainer(model=model, optimizer=optimizer, lr_scheduler=lr_scheduler, loss_criterion=loss_criterion, eval_criterion=eval_criterion,
                       loaders=loaders, tensorboard_formatter=tensorboard_formatter, resume=resume, pre_trained=pre_trained, **trainer_config)


class _ScalarLog:
    """SummaryWriter stand-in / tee: keeps (tag, value, iteration) and forwards to tensorboard when it is installed"""

    def __init__(self, log_dir):
        self.scalars = []
        try:
            from torch.utils.tensorboard import SummaryWriter
            self.tb = SummaryWriter(log_dir=log_dir)
        except Exception:                       # tensorboard not installed
            self.tb = None

    def add_scalar(self, tag, value, it):
        self.scalars.append((tag, float(value), int(it)))
        if self.tb is not None:
            self.tb.add_scalar(tag, value, it)

    def add_image(self, tag, image, it):
        if self.tb is not None:
            self.tb.add_image(tag, image, it)


class UNetTrainer:
    def __init__(self, model, optimizer, lr_scheduler, loss_criterion, eval_criterion, loaders, checkpoint_dir, max_num_epochs, max_num_iterations,
                 validate_after_iters=200, log_after_iters=100, validate_iters=None, num_iterations=1, num_epoch=0, eval_score_higher_is_better=True,
                 tensorboard_formatter=None, skip_train_validation=False, resume=None, pre_trained=None, **kwargs):
        self.model, self.optimizer, self.scheduler = model, optimizer, lr_scheduler
        self.loss_criterion, self.eval_criterion, self.loaders = loss_criterion, eval_criterion, loaders
        self.checkpoint_dir = checkpoint_dir
        self.max_num_epochs, self.max_num_iterations = max_num_epochs, max_num_iterations
        self.validate_after_iters, self.log_after_iters, self.validate_iters = validate_after_iters, log_after_iters, validate_iters
        self.eval_score_higher_is_better = eval_score_higher_is_better
        self.best_eval_score = float("-inf") if eval_score_higher_is_better else float("+inf")
        self.writer = _ScalarLog(os.path.join(checkpoint_dir, "logs", datetime.now().strftime("%Y-%m-%d_%H-%M-%S")))
        self.tensorboard_formatter = tensorboard_formatter            # None: no images are logged (the reference requires one)
        self.num_iterations, self.num_epochs = num_iterations, num_epoch
        self.skip_train_validation = skip_train_validation
        if resume is not None:
            state = utils.load_checkpoint(resume, self.model, self.optimizer)
            logger.info(f"Checkpoint loaded from '{resume}'. Epoch: {state['num_epochs']}.  Iteration: {state['num_iterations']}. "
                        f"Best val score: {state['best_eval_score']}.")
            self.best_eval_score, self.num_iterations, self.num_epochs = state["best_eval_score"], state["num_iterations"], state["num_epochs"]
            self.checkpoint_dir = os.path.split(resume)[0]
        elif pre_trained is not None:
            utils.load_checkpoint(pre_trained, self.model, None)
            if "checkpoint_dir" not in kwargs:
                self.checkpoint_dir = os.path.split(pre_trained)[0]

    @property
    def scalars(self):
        return self.writer.scalars

    def fit(self):
        for _ in range(self.num_epochs, self.max_num_epochs):
            if self.train():
                logger.info("Stopping criterion is satisfied. Finishing training")
                return
            self.num_epochs += 1
        logger.info(f"Reached maximum number of epochs: {self.max_num_epochs}. Finishing training...")

    def _net(self):
        return self.model.module if isinstance(self.model, nn.DataParallel) else self.model

    def train(self):
        """one epoch; True = stop now (trainer.py:166-240)"""
        train_losses, train_eval_scores = utils.RunningAverage(), utils.RunningAverage()
        self.model.train()
        for t in self.loaders["train"]:
            input, target, weight = self._split_training_batch(t)
            output, loss = self._forward_pass(input, target, weight)
            train_losses.update(loss.item(), self._batch_size(input))
            self.optimizer.zero_grad()
            loss.backward()
            self.optimizer.step()
            if self.num_iterations % self.validate_after_iters == 0:
                self.model.eval()
                eval_score = self.validate()
                self.model.train()
                if isinstance(self.scheduler, ReduceLROnPlateau):
                    self.scheduler.step(eval_score)
                elif self.scheduler is not None:
                    self.scheduler.step()
                self._log_lr()
                self._save_checkpoint(self._is_best_eval_score(eval_score))
            if self.num_iterations % self.log_after_iters == 0:
                if not self.skip_train_validation:
                    act = self._net().final_activation
                    eval_score = self.eval_criterion(act(output) if act is not None else output, target)
                    train_eval_scores.update(eval_score.item(), self._batch_size(input))
                logger.info(f"Training stats. Loss: {train_losses.avg}. Evaluation score: {train_eval_scores.avg}")
                self._log_stats("train", train_losses.avg, train_eval_scores.avg)
                self._log_images(input, target, output, "train_")
            if self.should_stop():
                return True
            self.num_iterations += 1
        return False

    def should_stop(self):
        if self.max_num_iterations < self.num_iterations:
            logger.info(f"Maximum number of iterations {self.max_num_iterations} exceeded.")
            return True
        if self.optimizer.param_groups[0]["lr"] < 1e-6:
            logger.info("Learning rate below the minimum 1e-06.")
            return True
        return False

    def validate(self):
        val_losses, val_scores = utils.RunningAverage(), utils.RunningAverage()
        with torch.no_grad():
            for i, t in enumerate(self.loaders["val"]):
                input, target, weight = self._split_training_batch(t)
                output, loss = self._forward_pass(input, target, weight)
                val_losses.update(loss.item(), self._batch_size(input))
                if i % 100 == 0:
                    self._log_images(input, target, output, "val_")
                val_scores.update(self.eval_criterion(output, target).item(), self._batch_size(input))
                if self.validate_iters is not None and self.validate_iters <= i:
                    break
            self._log_stats("val", val_losses.avg, val_scores.avg)
            logger.info(f"Validation finished. Loss: {val_losses.avg}. Evaluation score: {val_scores.avg}")
            return val_scores.avg

    def _split_training_batch(self, t):
        def to_gpu(v):
            if isinstance(v, (tuple, list)):
                return tuple(to_gpu(x) for x in v)
            return v.cuda(non_blocking=True)

        t = to_gpu(t)
        if len(t) == 2:
            return t[0], t[1], None
        return t[0], t[1], t[2]

    def _forward_pass(self, input, target, weight=None):
        output = self.model(input)
        loss = self.loss_criterion(output, target) if weight is None else self.loss_criterion(output, target, weight)
        return output, loss

    def _is_best_eval_score(self, eval_score):
        is_best = eval_score > self.best_eval_score if self.eval_score_higher_is_better else eval_score < self.best_eval_score
        if is_best:
            logger.info(f"Saving new best evaluation metric: {eval_score}")
            self.best_eval_score = eval_score
        return is_best

    def _save_checkpoint(self, is_best):
        utils.save_checkpoint({
            "num_epochs": self.num_epochs + 1,
            "num_iterations": self.num_iterations,
            "model_state_dict": self._net().state_dict(),
            "best_eval_score": self.best_eval_score,
            "optimizer_state_dict": self.optimizer.state_dict(),
        }, is_best, checkpoint_dir=self.checkpoint_dir)

    def _log_lr(self):
        self.writer.add_scalar("learning_rate", self.optimizer.param_groups[0]["lr"], self.num_iterations)

    def _log_stats(self, phase, loss_avg, eval_score_avg):
        self.writer.add_scalar(f"{phase}_loss_avg", loss_avg, self.num_iterations)
        self.writer.add_scalar(f"{phase}_eval_score_avg", eval_score_avg, self.num_iterations)

    def _log_images(self, input, target, prediction, prefix=""):
        if self.tensorboard_formatter is None or self.writer.tb is None:
            return
        act = self._net().final_activation
        if act is not None:
            prediction = act(prediction)
        for name, batch in (("inputs", input), ("targets", target), ("predictions", prediction)):
            parts = {f"{name}{i}": b for i, b in enumerate(batch)} if isinstance(batch, (list, tuple)) else {name: batch}
            for key, b in parts.items():
                for tag, image in self.tensorboard_formatter(key, b.data.cpu().numpy()):
                    self.writer.add_image(prefix + tag, image, self.num_iterations)

    @staticmethod
    def _batch_size(input):
        return input[0].size(0) if isinstance(input, (list, tuple)) else input.size(0)
