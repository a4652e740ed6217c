"""Mirror of trainer/MYtrainer.py:6-11: `CustomTrainer(transformers.Trainer).compute_loss`.
Also accepts the `num_items_in_batch` keyword that transformers >= 4.46 passes (SURVEY.md §8b version drift)."""
from transformers import Trainer


class CustomTrainer(Trainer):
    def compute_loss(self, model, inputs, return_outputs=False, **kwargs):
        labels = inputs.get("labels")  # noqa: F841  (kept: the reference reads it too)
        outputs = model(**inputs)
        loss = outputs["loss"]
        return (loss, outputs) if return_outputs else loss
