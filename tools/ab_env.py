"""Interleaved A/B of launcher tuning switches (environment variables read per launch) over every
MFMA GEMM of one train step, in ONE process.  usage: ab_env.py VAR=a,b[,c] [VAR2=...] [--only substr]"""
import os, sys, itertools
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vae_eeg_xai_amd import roofline as rl, _lib

args = [a for a in sys.argv[1:] if "=" in a]
only = None
if "--only" in sys.argv:
    only = sys.argv[sys.argv.index("--only") + 1]
names = [a.split("=")[0] for a in args]
vals = [a.split("=")[1].split(",") for a in args]
combos = list(itertools.product(*vals))
cases = rl.gemm_cases(512, bf16=("--bf16" in sys.argv))
print("kernel".ljust(22) + "".join(("/".join(c)).rjust(12) for c in combos), " (us; vars: " + ",".join(names) + ")")
tot = [0.0] * len(combos)
for label, macs, fn in cases:
    if only and only not in label:
        continue
    row = []
    for ci, c in enumerate(combos):
        for n, v in zip(names, c):
            _lib.setenv(n, v)
        t = rl._time_launch(fn, iters=10, warm=2, rounds=3, warm_ms=15.0)
        row.append(t * 1e6)
        tot[ci] += t * 1e6
    print(label.ljust(22) + "".join(f"{t:12.1f}" for t in row))
print("sum".ljust(22) + "".join(f"{t:12.1f}" for t in tot))
