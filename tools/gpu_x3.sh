#!/bin/bash
set -u
mkdir -p gpurun_out
timeout -k 10 400 python -u -m pytest tests/test_gpu_net.py -m gpu -x -q --timeout 200 --timeout-method thread -k "split_fp32 or forward_loss_and_every" > gpurun_out/x3_tests.log 2>&1
rc=$?; echo "pytest rc=$rc"; tail -n 15 gpurun_out/x3_tests.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 300 python -u bench.py --dtype f32x3 --no-cpu-baseline --no-attribution --no-roofline > gpurun_out/x3_bench.json 2>gpurun_out/x3_bench.err
echo "bench rc=$?"; cut -c1-400 gpurun_out/x3_bench.json
timeout -k 10 300 python -u bench.py --dtype f32 --no-cpu-baseline --no-attribution --no-roofline --no-bf16 > gpurun_out/x3_bench_f32.json 2>gpurun_out/x3_bench_f32.err
echo "bench rc=$?"; cut -c1-400 gpurun_out/x3_bench_f32.json
