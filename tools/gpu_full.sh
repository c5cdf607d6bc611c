#!/bin/bash
# Full GPU check of the tree: the gpu-marked tests, smoke(), and the default bench line.
set -u
OUT=gpurun_out; mkdir -p $OUT
timeout -k 10 1000 python -u -m pytest tests -m gpu -x -v --timeout 300 --timeout-method thread > $OUT/full_pytest.log 2>&1
rc=$?; echo "pytest rc=$rc"; tail -n 12 $OUT/full_pytest.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 300 python -u -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $OUT/full_smoke.log 2>&1
rc=$?; echo "smoke rc=$rc"; tail -n 3 $OUT/full_smoke.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 600 python -u bench.py > $OUT/full_bench.json 2> $OUT/full_bench.err
rc=$?; echo "bench rc=$rc"; tail -n 2 $OUT/full_bench.json | cut -c1-1500
