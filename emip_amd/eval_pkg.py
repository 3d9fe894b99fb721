"""Alias target of the reference's `eval` package (train.py:24 `import eval.metrics as Measure`): carries `metrics`."""
from . import eval_metrics as metrics  # noqa: F401
