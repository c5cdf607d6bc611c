#!/bin/bash
# GPU box: interleaved A/B of one environment switch on the whole train step (bench.py, 3 rounds each).
# usage: bash tools/ab_bench.sh VAR a b [bench args]
cd "${GRAFT_REPO_ROOT:-.}"
V=$1; A=$2; B=$3; shift 3
for r in 1 2 3; do
  for x in $A $B; do
    env $V=$x timeout -k 10 120 python bench.py --steps 100 --warmup 20 --no-cpu-baseline --no-roofline --no-attribution "$@" 2>/dev/null \
      | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$V=$x', round(d['ms_per_step'],4), 'ms/step')" || exit 1
  done
done
