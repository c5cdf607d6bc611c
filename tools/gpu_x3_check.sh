#!/bin/bash
# GPU box: correctness of the split-fp32 kernels, then the per-launch table of that engine and a short bench.
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
timeout -k 10 400 python -u -m pytest tests/test_gpu_kernels.py tests/test_gpu_net.py -x -q -m gpu -k "split_fp32" --timeout 120 --timeout-method thread > gpurun_out/x3_tests.log 2>&1 || { tail -40 gpurun_out/x3_tests.log; exit 1; }
tail -3 gpurun_out/x3_tests.log
timeout -k 10 200 python -u -m vae_eeg_xai_amd.roofline 512 f32x3 > gpurun_out/x3_layer_table.txt 2>&1 || { tail -30 gpurun_out/x3_layer_table.txt; exit 1; }
cat gpurun_out/x3_layer_table.txt
timeout -k 10 200 python -u bench.py --no-cpu-baseline --no-roofline --no-attribution --no-bf16 > gpurun_out/x3_bench.json 2> gpurun_out/x3_bench.err || { tail -30 gpurun_out/x3_bench.err; exit 1; }
cat gpurun_out/x3_bench.json | cut -c1-200
