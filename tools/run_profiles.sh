#!/bin/bash
# Runs on the GPU box (via gpurun): kernel-trace stats of bench.py + separate PMC passes over the GEMM launches.
set -e
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
R=${1:-r01}
O=gpurun_out/$R
mkdir -p $O
# the default bench line (split-fp32 engine; its f32_mfma / bf16 objects carry the other engines)
timeout -k 10 600 python bench.py --steps 50 --warmup 10 > $O/bench.json 2> $O/bench.err
timeout -k 10 120 python -m vae_eeg_xai_amd.roofline 512 > $O/layer_table.txt 2>/dev/null
timeout -k 10 120 python -m vae_eeg_xai_amd.roofline 512 f32x3 > $O/layer_table_x3.txt 2>/dev/null
timeout -k 10 300 python bench.py --dtype bf16 --steps 50 --warmup 10 --no-cpu-baseline > $O/bench_bf16.json 2> $O/bench_bf16.err
timeout -k 10 300 python bench.py --dtype f32 --steps 50 --warmup 10 --no-cpu-baseline --no-bf16 > $O/bench_f32_mfma.json 2> $O/bench_f32_mfma.err
timeout -s KILL 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline --no-attribution --no-bf16 > $O/trace.log 2>&1
# split-fp32 engine: counter passes over ITS launches (forward, input gradient and weight gradient on the split kernels)
timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch_x3 -- python tools/prof_gemms.py 512 2 f32x3 > $O/pmc_fetch_x3.log 2>&1
timeout -s KILL 120 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write_x3 -- python tools/prof_gemms.py 512 2 f32x3 > $O/pmc_write_x3.log 2>&1
timeout -s KILL 120 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $O/pmc_sq_x3 -- python tools/prof_gemms.py 512 2 f32x3 > $O/pmc_sq_x3.log 2>&1
timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -- python tools/prof_gemms.py 512 2 > $O/pmc_fetch.log 2>&1
timeout -s KILL 120 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -- python tools/prof_gemms.py 512 2 > $O/pmc_write.log 2>&1
timeout -s KILL 120 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $O/pmc_sq -- python tools/prof_gemms.py 512 2 > $O/pmc_sq.log 2>&1
# one train step with the side stream off: per-kernel times that no concurrent kernel stretches (fp32, bf16)
bash tools/prof_seq.sh $R/seq_x3 > $O/seq_x3.txt 2>&1
bash tools/prof_seq.sh $R/seq_f32 --dtype f32 > $O/seq_f32.txt 2>&1
bash tools/prof_seq.sh $R/seq_bf16 --dtype bf16 > $O/seq_bf16.txt 2>&1
# attribution only (batch 4096: 3 x top-1 unit, 3 x all 28 units)
timeout -s KILL 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/attr -- python tools/prof_attr.py > $O/attr.log 2>&1
[ -x tools/micro/mfma_issue ] && timeout -k 10 60 tools/micro/mfma_issue > $O/micro_mfma_issue.txt 2>&1
# HBM-bound kernels: achieved GB/s against 8 TB/s
timeout -k 10 200 python -m vae_eeg_xai_amd.roofline --hbm 512 > $O/hbm_table.txt 2>/dev/null
# bf16 engine: counter passes of their own (tools/pmc_bf16.sh writes $O/pmc_*_bf16)
bash tools/pmc_bf16.sh $R > $O/pmc_bf16.log 2>&1
# two ranks on the one GPU, collectives staged through the host (gloo): the N > 1 control flow of bench.py
VEEX_DIST_BACKEND=gloo VEEX_DIST_ONE_GPU=1 timeout -k 10 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 \
  --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --steps 10 --warmup 3 --no-roofline --no-bf16 > $O/dp2_rehearsal.json 2> $O/dp2_rehearsal.err
# latent sweep of configs[2] at its local batch (one GPU)
for L in 2 8 16 28 64 128; do
  timeout -k 10 200 python bench.py --batch 1024 --latent $L --steps 30 --warmup 10 --no-cpu-baseline --no-roofline --no-attribution --no-bf16 2>/dev/null >> $O/latent_sweep_b1024.jsonl
done
# the bench line over the batch size (fp32 and bf16): how the per-launch floors amortise
for B in 64 512 2048 4096; do
  timeout -k 10 200 python bench.py --batch $B --steps 30 --warmup 10 --no-cpu-baseline --no-roofline --no-attribution --no-bf16 2>/dev/null >> $O/bench_batches.jsonl
  timeout -k 10 200 python bench.py --dtype bf16 --batch $B --steps 30 --warmup 10 --no-cpu-baseline --no-roofline --no-attribution 2>/dev/null >> $O/bench_bf16_batches.jsonl
done
# the bf16 layer table again at batch 4096: how far the per-launch floors are from the roof once the launches are 8x longer
timeout -k 10 200 python -m vae_eeg_xai_amd.roofline 4096 bf16 > $O/layer_table_bf16_b4096.txt 2>/dev/null
echo PROFILES_DONE
