#!/bin/bash
set -u
OUT=gpurun_out; mkdir -p $OUT
timeout -k 10 600 python -u -m pytest tests/test_gpu_net.py -m gpu -x -q --timeout 200 --timeout-method thread -k "batch_statistics or alternate_semantics or probability" > $OUT/r3l_bn.log 2>&1
rc=$?; echo "bn pytest rc=$rc"; tail -n 25 $OUT/r3l_bn.log
[ $rc -eq 0 ] || [ $rc -eq 1 ] || exit $rc
timeout -k 10 900 python -u -m pytest tests/test_gpu_dp.py tests/test_gpu_driver.py -m gpu -x -q --timeout 400 --timeout-method thread > $OUT/r3l_pytest.log 2>&1
rc=$?; echo "pytest rc=$rc"; tail -n 15 $OUT/r3l_pytest.log
