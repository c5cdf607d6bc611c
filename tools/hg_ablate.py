"""Ablation of the halo-tile kernel in a `make TIMING=1` build (VEEX_HG_DBG drops parts of the kernel; results are
wrong by construction).  usage: python tools/hg_ablate.py batch label G bands CH NT"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from vae_eeg_xai_amd import _lib, roofline as rl  # noqa: E402
from hg_sweep import timed  # noqa: E402

B, want = int(sys.argv[1]), sys.argv[2]
_lib.setenv("VEEX_HG_PLAN", _lib.hg_plan_code(*[int(v) for v in sys.argv[3:7]]))
for label, macs, fn in rl.gemm_cases(B, True):
    if label != want:
        continue
    row = []
    for dbg, name in ((0, "full"), (1, "no mfma"), (2, "no epilogue"), (4, "no staging"), (8, "no weight chunks"), (16, "no barriers"),
                      (3, "no mfma, no epilogue"), (7, "no mfma/epi/staging"), (15, "only barriers + tables"), (31, "launch + tables"),
                      (25, "staging + epilogue only"), (63, "launch, no tables"), (64, "return at once")):
        _lib.setenv("VEEX_HG_DBG", dbg)
        row.append(f"{name}: {timed(fn):.1f}")
    print(label, " | ".join(row), flush=True)
