#!/bin/bash
# GPU box: HBM/fabric fetch bytes per MFMA launch (FETCH_SIZE pass over tools/prof_gemms.py) under different settings of
# environment switches.  usage: bash tools/pmc_fetch_ab.sh tag "VAR=a VAR2=b" "VAR=c" ...
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
T=$1; shift
i=0
for setting in "$@"; do
  O=gpurun_out/$T/$i
  mkdir -p $O
  ( export $setting; timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O -- python tools/prof_gemms.py 512 2 > $O/log.txt 2>&1 ) || { tail $O/log.txt; exit 1; }
  python - "$O" "$setting" <<'PY'
import csv, glob, sys, collections
o, setting = sys.argv[1], sys.argv[2]
order = open(o + "/log.txt").read().split("launch order:")[1].split()[0].split(",")
f = max(glob.glob(o + "/*/*counter_collection.csv"))
d = collections.OrderedDict()
for r in csv.DictReader(open(f)):
    if "gather_gemm" in r["Kernel_Name"] or "wgrad_" in r["Kernel_Name"]:
        d[int(r["Dispatch_Id"])] = float(r["Counter_Value"])
v = list(d.values())
reps = len(v) // len(order)
print(setting)
print("  " + "  ".join(f"{n.replace('enc.','').replace('dec.','')}={v[reps*i+reps-1]*2048/1e6:.0f}" for i, n in enumerate(order)), "(MB)")
PY
  i=$((i+1))
done
