"""tensorflow.python: only ops.numpy_ops.np_config exists (reference train.py:15)."""
