#!/bin/bash
set -u
OUT=gpurun_out; mkdir -p $OUT
export PYTHONPATH=tools
rc=0

[ $rc -eq 0 ] || exit $rc
: > $OUT/r3g_stagger.txt
for cfg in "dec.deconv4.fwd 1 1 64 1" "dec.deconv4.dgrad 1 1 64 1" "dec.deconv2.fwd 2 1 128 1" "dec.deconv2.dgrad 4 1 64 1" "dec.deconv5.fwd 1 2 64 1" "dec.deconv3.fwd 2 1 128 1" "enc.conv2.dgrad 1 1 128 1"; do
  timeout -k 10 120 python -u tools/hg_stagger.py 512 $cfg >> $OUT/r3g_stagger.txt 2>&1 || { tail $OUT/r3g_stagger.txt; exit 1; }
done
grep -v amdgpu.ids $OUT/r3g_stagger.txt
timeout -k 10 200 python -u tools/try_microbatch.py bf16 512 && timeout -k 10 200 python -u tools/try_microbatch.py f32 512
