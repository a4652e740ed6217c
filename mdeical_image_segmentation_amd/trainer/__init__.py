"""Mirror of the reference's `trainer` package (trainer/__init__.py:1-3)."""
from . import MYtrainer, metrcis  # noqa: F401
from .MYtrainer import CustomTrainer  # noqa: F401
from .metrcis import compute_metrics  # noqa: F401
