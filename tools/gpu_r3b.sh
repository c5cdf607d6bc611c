#!/bin/bash
# round 3: halo-tile bf16 kernel -- parity against the general kernel, then the per-launch table with and without it
set -u
OUT=gpurun_out; mkdir -p $OUT
timeout -k 10 600 python -u -m pytest tests/test_gpu_bf16.py tests/test_gpu_net.py -m gpu -x -q --timeout 300 --timeout-method thread -k "halo or workspace or graphed or bf16" > $OUT/r3b_pytest.log 2>&1
rc=$?; echo "pytest rc=$rc"; tail -n 25 $OUT/r3b_pytest.log
[ $rc -eq 0 ] || exit $rc
VEEX_HALO=0 timeout -k 10 200 python -m vae_eeg_xai_amd.roofline 512 bf16 > $OUT/r3b_table_bf16_general.txt 2>&1 || { tail $OUT/r3b_table_bf16_general.txt; exit 1; }
timeout -k 10 200 python -m vae_eeg_xai_amd.roofline 512 bf16 > $OUT/r3b_table_bf16_halo.txt 2>&1 || { tail $OUT/r3b_table_bf16_halo.txt; exit 1; }
paste -d'\n' $OUT/r3b_table_bf16_general.txt $OUT/r3b_table_bf16_halo.txt | cut -c1-150
