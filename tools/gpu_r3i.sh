#!/bin/bash
set -u
OUT=gpurun_out; mkdir -p $OUT
timeout -k 10 1100 python -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread > $OUT/r3i_pytest.log 2>&1
rc=$?; echo "pytest rc=$rc"; tail -n 15 $OUT/r3i_pytest.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 200 python -m vae_eeg_xai_amd.roofline 512 bf16 > $OUT/r3i_table_bf16.txt 2>&1 || { tail $OUT/r3i_table_bf16.txt; exit 1; }
grep -v amdgpu.ids $OUT/r3i_table_bf16.txt
timeout -k 10 400 python bench.py --steps 30 --warmup 10 --no-cpu-baseline --no-attribution > $OUT/r3i_bench.json 2> $OUT/r3i_bench.err || { tail $OUT/r3i_bench.err; exit 1; }
cat $OUT/r3i_bench.json
