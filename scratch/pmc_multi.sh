#!/bin/bash
# usage: scratch/pmc_multi.sh <outdir> "<counters pass1>" "<counters pass2>" ... -- program args
set -e
out=$1; shift
passes=()
while [ "$1" != "--" ]; do passes+=("$1"); shift; done
shift
export TMPDIR=/tmp
root=$(pwd)
i=0
for p in "${passes[@]}"; do
  mkdir -p gpurun_out/$out/p$i
  (cd /tmp && rocprofv3 --pmc $p --output-format csv -d $root/gpurun_out/$out/p$i -- "$@" > $root/gpurun_out/$out/p$i/run.log 2>&1) || { echo "pass $i failed"; tail -5 gpurun_out/$out/p$i/run.log; }
  i=$((i+1))
done
python3 - <<PY
import csv, glob, collections
agg = collections.defaultdict(list)
for f in glob.glob("gpurun_out/$out/p*/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0][-60:]
        agg[(k, r["Counter_Name"])].append(float(r["Counter_Value"]))
for (k, c), v in sorted(agg.items()):
    if len(v) >= 20: print("%-62s %-22s n=%4d mean=%14.1f" % (k, c, len(v), sum(v)/len(v)))
PY
