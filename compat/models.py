"""``from models import *`` shim: put this directory on sys.path ahead of the reference's own
models.py and train.py / analysis.py / signal_reconstruction.py pick up the MI355X implementation
(reference train.py:18, analysis.py:3, signal_reconstruction.py:7)."""
import os as _os, sys as _sys
_sys.path.insert(0, _os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))))
from vae_eeg_xai_amd.models import *      # noqa: F401,F403
from vae_eeg_xai_amd.models import tf, keras  # noqa: F401
