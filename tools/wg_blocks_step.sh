#!/bin/bash
# (make -C vae_eeg_xai_amd/csrc timing first)  GPU box, tuning build: the row-split target of the weight-gradient GEMMs (VEEX_WG_BLOCKS) on the whole split-fp32 train step
cd "${GRAFT_REPO_ROOT:-.}"
cp vae_eeg_xai_amd/libveex_timing.so vae_eeg_xai_amd/libveex.so
mkdir -p gpurun_out; : > gpurun_out/wg_blocks_step.txt
for rep in 1 2; do
for T in 384 512 640 768 1024; do
  VEEX_WG_BLOCKS=$T timeout -k 10 200 python -u bench.py --no-cpu-baseline --no-roofline --no-attribution --no-bf16 2>/dev/null | python -c "import json,sys; print('VEEX_WG_BLOCKS=$T step ms', round(json.loads(sys.stdin.read().strip().splitlines()[-1])['ms_per_step'],4))" >> gpurun_out/wg_blocks_step.txt || exit 1
done
done
cat gpurun_out/wg_blocks_step.txt
