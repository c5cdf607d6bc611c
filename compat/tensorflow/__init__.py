"""``tensorflow`` name for the reference's drivers: tf.config.list_physical_devices (train.py:248),
tf.keras.backend.clear_session (train.py:207).  No TensorFlow is involved."""
import os as _os, sys as _sys
_root = _os.path.abspath(__file__)
for _ in range(3):
    _root = _os.path.dirname(_root)
if _root not in _sys.path:
    _sys.path.insert(0, _root)
from vae_eeg_xai_amd.keras_compat import tf as _tf
config = _tf.config
keras = _tf.keras
