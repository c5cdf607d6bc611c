"""keras.src: the private path the reference imports EarlyStopping from (train.py:10)."""
from . import callbacks  # noqa: F401
