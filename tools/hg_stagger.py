"""Stagger sweep of the halo-tile kernel: usage: python tools/hg_stagger.py batch label G bands CH NT"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from vae_eeg_xai_amd import _lib, roofline as rl  # noqa: E402
from hg_sweep import timed  # noqa: E402

B, want = int(sys.argv[1]), sys.argv[2]
torch.zeros(1, device="cuda")
for k, v in zip(("VEEX_HG_G", "VEEX_HG_BANDS", "VEEX_HG_CH", "VEEX_HG_NT"), sys.argv[3:7]):
    _lib.setenv(k, v)
for label, macs, fn in rl.gemm_cases(B, True):
    if label != want:
        continue
    row = []
    for st in (0, 1, 2, 3, 4, 6, 8):
        _lib.setenv("VEEX_HG_STAGGER", st)
        row.append(f"stagger {st}: {timed(fn):.1f}")
    print(label, sys.argv[3:7], " | ".join(row), flush=True)
