#!/bin/bash
# (make -C vae_eeg_xai_amd/csrc timing first)  GPU box: bf16 tests with the shipped library, then (tuning build) the transposing-read bf16 weight-gradient kernel against
# the copy kernel: per-launch table rows and the bf16 step, over its chunk depth and the row-split target.
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
timeout -k 10 500 python -u -m pytest tests/test_gpu_bf16.py -x -q -m gpu --timeout 200 --timeout-method thread > gpurun_out/wgbt_tests.log 2>&1 || { tail -40 gpurun_out/wgbt_tests.log; exit 1; }
tail -2 gpurun_out/wgbt_tests.log
cp vae_eeg_xai_amd/libveex_timing.so vae_eeg_xai_amd/libveex.so
: > gpurun_out/wgbt.txt
run() {
  echo "== $*" >> gpurun_out/wgbt.txt
  env "$@" timeout -k 10 200 python -u -m vae_eeg_xai_amd.roofline 512 bf16 2>/dev/null | grep wgrad | cut -c1-60 >> gpurun_out/wgbt.txt || exit 1
  env "$@" timeout -k 10 200 python -u bench.py --dtype bf16 --no-cpu-baseline --no-roofline --no-attribution 2>/dev/null | python -c "import json,sys; print('step ms', json.loads(sys.stdin.read().strip().splitlines()[-1])['ms_per_step'])" >> gpurun_out/wgbt.txt || exit 1
}
run VEEX_WGBT=0
run VEEX_WGBT=1 VEEX_WGBT_BK=32 VEEX_WG_BLOCKS=256
run VEEX_WGBT=1 VEEX_WGBT_BK=32 VEEX_WG_BLOCKS=512
run VEEX_WGBT=1 VEEX_WGBT_BK=32 VEEX_WG_BLOCKS=768
run VEEX_WGBT=1 VEEX_WGBT_BK=64 VEEX_WG_BLOCKS=256
run VEEX_WGBT=1 VEEX_WGBT_BK=64 VEEX_WG_BLOCKS=512
cat gpurun_out/wgbt.txt
