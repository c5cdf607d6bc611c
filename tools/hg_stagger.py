"""Stagger sweep of the halo-tile kernel: usage: python tools/hg_stagger.py batch label G bands CH NT"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from vae_eeg_xai_amd import _lib, roofline as rl  # noqa: E402
from hg_sweep import timed  # noqa: E402

B, want = int(sys.argv[1]), sys.argv[2]
torch.zeros(1, device="cuda")
_lib.setenv("VEEX_HG_PLAN", _lib.hg_plan_code(*[int(v) for v in sys.argv[3:7]]))
for label, macs, fn in rl.gemm_cases(B, True):
    if label != want:
        continue
    row = []
    for st in (0, 1, 2, 3, 4, 6, 8):
        _lib.setenv("VEEX_HG_STAGGER", st)
        row.append(f"stagger {st}: {timed(fn):.1f}")
    print(label, sys.argv[3:7], " | ".join(row), flush=True)
