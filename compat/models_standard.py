"""``from models_standard import *`` shim (reference train_standard.py, analysis_standard.py)."""
import os as _os, sys as _sys
_sys.path.insert(0, _os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))))
from vae_eeg_xai_amd.models_standard import *      # noqa: F401,F403
from vae_eeg_xai_amd.models_standard import tf, keras  # noqa: F401
