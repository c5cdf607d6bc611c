#!/bin/bash
set -u
OUT=gpurun_out; mkdir -p $OUT
timeout -k 10 600 python -u -m pytest tests/test_gpu_bf16.py tests/test_gpu_net.py -m gpu -x -q --timeout 300 --timeout-method thread -k "halo or workspace" > $OUT/r3c_pytest.log 2>&1
rc=$?; echo "pytest rc=$rc"; tail -n 25 $OUT/r3c_pytest.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 900 python -u tools/hg_sweep.py 512 > $OUT/r3c_sweep.txt 2>&1 || { tail $OUT/r3c_sweep.txt; exit 1; }
cat $OUT/r3c_sweep.txt
