#!/bin/bash
# (make -C vae_eeg_xai_amd/csrc timing first)  GPU box, tuning build: the row-split targets of the split weight-gradient GEMMs
# (VEEX_WG_BLOCKS; VEEX_WG_BLOCKS64 was a target of its own for the 128 x 64 tile, tried and not shipped) on the whole split-fp32 train step
cd "${GRAFT_REPO_ROOT:-.}"
cp vae_eeg_xai_amd/libveex_timing.so vae_eeg_xai_amd/libveex.so
mkdir -p gpurun_out; : > gpurun_out/wg_blocks_step.txt
for rep in 1 2; do
for T in "512 512" "512 768" "512 1024" "768 768" "512 1536"; do
  set -- $T
  VEEX_WG_BLOCKS=$1 VEEX_WG_BLOCKS64=$2 timeout -k 10 200 python -u bench.py --no-cpu-baseline --no-roofline --no-attribution --no-bf16 2>/dev/null | python -c "import json,sys; print('VEEX_WG_BLOCKS=$1 VEEX_WG_BLOCKS64=$2 step ms', round(json.loads(sys.stdin.read().strip().splitlines()[-1])['ms_per_step'],4))" >> gpurun_out/wg_blocks_step.txt || exit 1
done
done
cat gpurun_out/wg_blocks_step.txt
