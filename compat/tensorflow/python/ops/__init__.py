"""tensorflow.python.ops: only numpy_ops.np_config exists (reference train.py:15)."""
