#!/bin/bash
# round 3, first GPU call: the -m gpu suite on the host-side fixes, then the bf16 per-launch table as the baseline
set -u
OUT=gpurun_out; mkdir -p $OUT
timeout -k 10 1000 python -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread > $OUT/r3a_pytest.log 2>&1
rc=$?; echo "pytest rc=$rc"; tail -n 15 $OUT/r3a_pytest.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 200 python -m vae_eeg_xai_amd.roofline 512 bf16 > $OUT/r3a_table_bf16.txt 2>&1 || { tail $OUT/r3a_table_bf16.txt; exit 1; }
cat $OUT/r3a_table_bf16.txt
timeout -k 10 200 python bench.py --dtype bf16 --steps 30 --warmup 10 --no-cpu-baseline --no-attribution > $OUT/r3a_bench_bf16.json 2> $OUT/r3a_bench_bf16.err || { tail $OUT/r3a_bench_bf16.err; exit 1; }
cat $OUT/r3a_bench_bf16.json
