"""Build-time guard (no GPU): the K loops of the hot kernels must not contain readfirstlane "waterfall" loops.
hipcc wraps a buffer load in such a loop when the descriptor or the scalar offset it was given is not provably
wave-uniform; it happened twice this round (a per-thread select of two descriptors; block coordinates computed
through an integer division) and cost 4-12 % of a train step each time without failing any test."""
import os
import re
import shutil
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "vae_eeg_xai_amd", "csrc")
sys.path.insert(0, os.path.join(ROOT, "tools"))

# file -> kernels that may keep their (known, cold: prologue of the spatial-affine variants) waterfall loops
HOT = {"gather_gemm_v3.hip": (), "gather_gemm_v4.hip": (), "gather_gemm_bf16.hip": (), "wgrad_gemm_bf16.hip": (),
       "wgrad_gemm_v3.hip": ("ELb0ELi2E",)}


def _waterfalls(asm_text):
    lines = asm_text.split("\n")
    kernel, found = None, {}
    for i, l in enumerate(lines):
        m = re.match(r"^(_Z\w+):", l)
        if m:
            kernel = m.group(1)
        if "s_cbranch_execnz" in l:
            tgt = l.split()[-1]
            j = i
            while j > 0 and i - j < 24 and not lines[j].startswith(tgt + ":"):
                j -= 1
            body = lines[j:i]
            if lines[j].startswith(tgt + ":") and any("v_readfirstlane" in b for b in body) and \
               any(re.search(r"\b(buffer|global)_", b) for b in body):
                found[kernel] = found.get(kernel, 0) + 1
    return found


@pytest.mark.skipif(shutil.which("hipcc") is None, reason="hipcc not on PATH")
@pytest.mark.parametrize("src", sorted(HOT))
def test_no_waterfall_loops_in_hot_kernels(src, tmp_path):
    out = tmp_path / (src + ".s")
    subprocess.run(["hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-S", "--cuda-device-only", "-o", str(out), src],
                   cwd=CSRC, check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    found = _waterfalls(out.read_text())
    bad = {k: v for k, v in found.items() if not any(tag in k for tag in HOT[src])}
    assert not bad, f"buffer loads inside readfirstlane loops: {bad}"
