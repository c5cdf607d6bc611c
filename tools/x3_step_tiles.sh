#!/bin/bash
# (make -C vae_eeg_xai_amd/csrc timing first)  GPU box, tuning build: one layer's tile of the split gather kernel changed INSIDE
# the whole train step (VEEX_X3_T<row of X3_KNOWN>=<cfg>): the input-gradient launches share the CUs with the weight gradients
# of the side stream, where the tile that is best on its own need not be.  usage: bash tools/x3_step_tiles.sh "A=1 B=2" "C=3" ...
cd "${GRAFT_REPO_ROOT:-.}"
cp vae_eeg_xai_amd/libveex_timing.so vae_eeg_xai_amd/libveex.so
mkdir -p gpurun_out; : > gpurun_out/x3_step_tiles.txt
for rep in 1 2; do
for S in "VEEX_NOP=0" "$@"; do
  env $S timeout -k 10 200 python -u bench.py --no-cpu-baseline --no-roofline --no-attribution --no-bf16 2>/dev/null | python -c "import json,sys; print('$S step ms', round(json.loads(sys.stdin.read().strip().splitlines()[-1])['ms_per_step'],4))" >> gpurun_out/x3_step_tiles.txt || exit 1
done
done
cat gpurun_out/x3_step_tiles.txt
