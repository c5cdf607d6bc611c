#!/bin/bash
set -u
OUT=gpurun_out; mkdir -p $OUT
export PYTHONPATH=tools
timeout -k 10 300 python -u -m pytest tests/test_gpu_bf16.py -m gpu -x -q --timeout 300 --timeout-method thread -k "halo" > $OUT/r3h_pytest.log 2>&1
rc=$?; echo "pytest rc=$rc"; tail -n 5 $OUT/r3h_pytest.log
[ $rc -eq 0 ] || exit $rc
: > $OUT/r3h_stagger.txt
for cfg in "dec.deconv4.fwd 1 1 64 1" "dec.deconv4.dgrad 1 1 64 1" "dec.deconv2.fwd 2 1 128 1" "dec.deconv2.dgrad 4 1 64 1" "dec.deconv5.fwd 1 2 64 1"; do
  timeout -k 10 120 python -u tools/hg_stagger.py 512 $cfg >> $OUT/r3h_stagger.txt 2>&1 || { tail $OUT/r3h_stagger.txt; exit 1; }
done
grep -v amdgpu.ids $OUT/r3h_stagger.txt
VEEX_HG_STAGGER=0 timeout -k 10 900 python -u tools/hg_sweep.py 512 > $OUT/r3h_sweep.txt 2>&1 || { tail $OUT/r3h_sweep.txt; exit 1; }
grep -v amdgpu.ids $OUT/r3h_sweep.txt | cut -c1-260
