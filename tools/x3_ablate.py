"""Split-fp32 gather kernel with one ingredient removed at a time (VEEX_X3_DBG of a `make TIMING=1` build; the results of
such a launch are wrong by construction): what each ingredient costs in the full loop.
usage: make -C vae_eeg_xai_amd/csrc timing; VEEX_LIB=vae_eeg_xai_amd/libveex_timing.so python tools/x3_ablate.py [batch]"""
import ctypes as C
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
if os.environ.get('VEEX_LIB'):
    import shutil; shutil.copy(os.environ['VEEX_LIB'], os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'vae_eeg_xai_amd', 'libveex.so'))
from vae_eeg_xai_amd import _lib, roofline as rl  # noqa: E402
from vae_eeg_xai_amd._lib import lib, check  # noqa: E402


def timed(fn, reps=20, groups=3):
    ts = []
    for _ in range(groups):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        fn(); torch.cuda.synchronize()
        a.record()
        for _ in range(reps):
            fn()
        b.record(); torch.cuda.synchronize()
        ts.append(a.elapsed_time(b) / reps * 1e3)
    return sorted(ts)[len(ts) // 2]


VARIANTS = [(0, "full"), (1, "no global loads"), (2, "no split arithmetic"), (4, "no LDS stores"), (8, "no MFMAs"),
            (16, "no operand reads"), (3, "no loads, no split"), (7, "no loads, split, stores"), (24, "no reads, no MFMAs")]


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 512
    torch.zeros(1, device="cuda")
    print("us per launch, batch %d: " % B + " | ".join(n for _, n in VARIANTS))
    for label, macs, fn in rl.gemm_cases(B, False, x3=True):
        if label.endswith("wgrad"):
            continue
        row = []
        for dbg, _ in VARIANTS:
            _lib.setenv("VEEX_X3_DBG", dbg if dbg else None)
            row.append(timed(fn))
        _lib.setenv("VEEX_X3_DBG", None)
        print(f"{label:20s} " + " | ".join(f"{t:7.1f}" for t in row), flush=True)


if __name__ == "__main__":
    main()
