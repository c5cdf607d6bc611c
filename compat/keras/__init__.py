"""``keras`` name for the reference's drivers (train.py:9-10): only the few symbols they import exist.
No Keras is involved; see vae_eeg_xai_amd/keras_compat.py."""
import os as _os, sys as _sys
_root = _os.path.abspath(__file__)
for _ in range(3):
    _root = _os.path.dirname(_root)
if _root not in _sys.path:
    _sys.path.insert(0, _root)
from vae_eeg_xai_amd.keras_compat import tf as _tf
backend = _tf.keras.backend
from . import callbacks, optimizers  # noqa: E402,F401
