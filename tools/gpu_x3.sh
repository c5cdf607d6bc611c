#!/bin/bash
set -u
mkdir -p gpurun_out
timeout -k 10 600 python -u -m pytest tests/test_gpu_net.py tests/test_gpu_kernels.py -m gpu -x -q --timeout 300 --timeout-method thread -k "split_fp32" > gpurun_out/x3_tests.log 2>&1
rc=$?; echo "pytest rc=$rc"; tail -n 25 gpurun_out/x3_tests.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 600 python -u bench.py > gpurun_out/x3_bench.json 2>gpurun_out/x3_bench.err
echo "bench rc=$?"; cat gpurun_out/x3_bench.json
