#!/bin/bash
set -u
OUT=gpurun_out; mkdir -p $OUT
export PYTHONPATH=tools
timeout -k 10 900 python -u tools/hg_sweep.py 512 > $OUT/r3e_sweep.txt 2>&1 || { tail $OUT/r3e_sweep.txt; exit 1; }
cat $OUT/r3e_sweep.txt | cut -c1-330
cd vae_eeg_xai_amd/csrc && rm -f *.o && make TIMING=1 -j16 > /dev/null 2>&1; cd ../..
timeout -k 10 120 python -u tools/hg_ablate.py 512 dec.deconv4.fwd 1 1 64 1 > $OUT/r3e_ablate.txt 2>&1 || { tail $OUT/r3e_ablate.txt; exit 1; }
timeout -k 10 120 python -u tools/hg_ablate.py 512 dec.deconv2.fwd 4 1 256 1 >> $OUT/r3e_ablate.txt 2>&1 || { tail $OUT/r3e_ablate.txt; exit 1; }
timeout -k 10 120 python -u tools/hg_ablate.py 512 dec.deconv4.dgrad 1 1 64 1 >> $OUT/r3e_ablate.txt 2>&1 || { tail $OUT/r3e_ablate.txt; exit 1; }
timeout -k 10 120 python -u tools/hg_ablate.py 512 dec.deconv5.fwd 1 2 64 1 >> $OUT/r3e_ablate.txt 2>&1 || { tail $OUT/r3e_ablate.txt; exit 1; }
cat $OUT/r3e_ablate.txt
