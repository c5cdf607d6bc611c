#!/bin/bash
set -u
OUT=gpurun_out; mkdir -p $OUT
timeout -k 10 600 python -u -m pytest tests/test_gpu_net.py -m gpu -q --timeout 200 --timeout-method thread -k "batch_statistics or alternate_semantics or probability" > $OUT/r3m_bn.log 2>&1
rc=$?; echo "bn pytest rc=$rc"; tail -n 40 $OUT/r3m_bn.log
