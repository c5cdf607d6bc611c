"""``from keras.src.callbacks import EarlyStopping`` (reference train.py:10)."""
import os as _os, sys as _sys
_root = _os.path.abspath(__file__)
for _ in range(4):
    _root = _os.path.dirname(_root)
if _root not in _sys.path:
    _sys.path.insert(0, _root)
from vae_eeg_xai_amd.keras_compat import EarlyStopping, History  # noqa: F401
