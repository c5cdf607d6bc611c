#!/bin/bash
# GPU box: per-kernel time of one train step with the side stream off (VEEX_OVERLAP=0), so that no kernel's
# duration is stretched by a concurrent one.  usage: bash tools/prof_seq.sh <tag> [bench args]
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp VEEX_OVERLAP=0
T=${1:-seq}; shift
O=gpurun_out/$T
mkdir -p $O
timeout -s KILL 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline --no-attribution --no-bf16 "$@" > $O/trace.log 2>&1 || { tail $O/trace.log; exit 1; }
python - "$O" <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/trace/**/*kernel_stats.csv", recursive=True)[0]
tot = 0.0
for r in csv.DictReader(open(f)):
    per_step = float(r["TotalDurationNs"]) / 25e3
    tot += per_step
    print(f'{r["Name"][:84]:84s} calls/step={int(r["Calls"]) / 25:5.1f} avg_us={float(r["AverageNs"]) / 1e3:8.1f} us/step={per_step:8.1f}')
print("total us/step", round(tot, 1))
PY
