#!/bin/bash
# One GPU-box call of round 2: the -m gpu suite, the two-rank rehearsal of bench.py on the one GPU (gloo,
# host-staged collectives), and the default bench line.  Each GPU step has its own time limit; after a step that
# did not end by itself (timeout / signal / abort) nothing more is started.
# usage: tools/gpu_r2.sh TAG [pytest args...]
set -u
TAG=${1:-r2}; shift || true
OUT=gpurun_out
mkdir -p $OUT
ok_to_go_on() { [ "$1" -eq 0 ] || [ "$1" -eq 1 ]; }   # pytest: 1 = some tests failed (assertions), everything else stops the call

timeout -k 10 1000 python -u -m pytest tests -m gpu -v --timeout 300 --timeout-method thread "$@" > $OUT/${TAG}_pytest.log 2>&1
rc=$?; echo "pytest rc=$rc"; tail -n 25 $OUT/${TAG}_pytest.log
ok_to_go_on $rc || exit $rc

VEEX_DIST_BACKEND=gloo VEEX_DIST_ONE_GPU=1 timeout -k 10 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 \
  --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --steps 10 --warmup 3 --no-roofline \
  > $OUT/${TAG}_dp2.json 2> $OUT/${TAG}_dp2.err
rc2=$?; echo "dp2 rehearsal rc=$rc2"; tail -n 5 $OUT/${TAG}_dp2.err; cat $OUT/${TAG}_dp2.json
[ $rc2 -eq 0 ] || exit $rc2

timeout -k 10 600 python bench.py > $OUT/${TAG}_bench.json 2> $OUT/${TAG}_bench.err
rc3=$?; echo "bench rc=$rc3"; tail -n 3 $OUT/${TAG}_bench.err; cat $OUT/${TAG}_bench.json
[ $rc3 -eq 0 ] || exit $rc3
exit $rc
