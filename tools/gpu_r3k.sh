#!/bin/bash
set -u
OUT=gpurun_out; mkdir -p $OUT
timeout -k 10 900 python -u -m pytest tests/test_gpu_dp.py tests/test_gpu_driver.py -m gpu -x -q --timeout 400 --timeout-method thread > $OUT/r3k_pytest.log 2>&1
rc=$?; echo "pytest rc=$rc"; tail -n 15 $OUT/r3k_pytest.log
