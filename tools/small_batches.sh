#!/bin/bash
# GPU box: ms per step of the split-fp32 and the fp32-MFMA engine at small batches (bench.py --batch B --steps 40).
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out; : > gpurun_out/small_batches.txt
for B in 32 64 128 256 1024; do
  for E in f32 f32x3; do
    timeout -k 10 120 python -u bench.py --batch $B --steps 40 --warmup 10 --dtype $E --no-cpu-baseline --no-roofline --no-attribution --no-bf16 > gpurun_out/sb.json 2> gpurun_out/sb.err || { tail gpurun_out/sb.err; exit 1; }
    python - $B $E <<'PY' >> gpurun_out/small_batches.txt
import json, sys
d = json.loads(open("gpurun_out/sb.json").read().strip().splitlines()[-1])
print(f"batch {sys.argv[1]} engine {sys.argv[2]}: {d['ms_per_step']:.3f} ms")
PY
  done
done
cat gpurun_out/small_batches.txt
