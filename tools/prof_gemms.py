"""Launch every MFMA GEMM of one train step (batch 512) a few times through the operator-level
C-ABI -- the target program of the rocprofv3 --pmc passes (profiles/README.md)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vae_eeg_xai_amd import roofline as rl

B = int(sys.argv[1]) if len(sys.argv) > 1 else 512
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
eng = sys.argv[3] if len(sys.argv) > 3 else "f32"      # f32 | f32x3 | bf16
cases = rl.gemm_cases(B, bf16=(eng == "bf16"), x3=(eng == "f32x3"))
for label, macs, fn in cases:
    for _ in range(reps):
        fn()
torch.cuda.synchronize()
print("launch order:", ",".join(l for l, _, _ in cases))
