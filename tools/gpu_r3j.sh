#!/bin/bash
set -u
OUT=gpurun_out; mkdir -p $OUT
timeout -k 10 600 python -u -m pytest tests/test_gpu_driver.py tests/test_gpu_net.py tests/test_dataio.py -m gpu -x -q --timeout 300 --timeout-method thread > $OUT/r3j_pytest.log 2>&1
rc=$?; echo "pytest rc=$rc"; tail -n 15 $OUT/r3j_pytest.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 400 python -u tools/fit_throughput.py f32 > $OUT/r3j_fit_f32.jsonl 2> $OUT/r3j_fit.err || { tail $OUT/r3j_fit.err; exit 1; }
cat $OUT/r3j_fit_f32.jsonl
