"""tensorflow.python.ops.numpy_ops: np_config only (reference train.py:15)."""
from . import np_config  # noqa: F401
