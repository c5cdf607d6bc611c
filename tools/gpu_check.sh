#!/bin/bash
# GPU box: parity tests, per-launch table, bench line. Usage: bash tools/gpu_check.sh <tag>
cd "${GRAFT_REPO_ROOT:-.}"
T=${1:-chk}
mkdir -p gpurun_out
timeout -k 10 300 python -u -m pytest tests -m gpu -x -q --timeout 120 --timeout-method thread > gpurun_out/${T}_pytest.log 2>&1 || { tail -30 gpurun_out/${T}_pytest.log; echo TESTS_FAILED; exit 1; }
tail -2 gpurun_out/${T}_pytest.log
timeout -k 10 200 python -m vae_eeg_xai_amd.roofline 512 > gpurun_out/${T}_table.txt 2>&1 || { tail gpurun_out/${T}_table.txt; exit 1; }
cat gpurun_out/${T}_table.txt
timeout -k 10 300 python bench.py --steps 50 --warmup 10 --no-cpu-baseline > gpurun_out/${T}_bench.json 2> gpurun_out/${T}_bench.err || { tail gpurun_out/${T}_bench.err; exit 1; }
cat gpurun_out/${T}_bench.json
