#!/bin/bash
export TMPDIR=/tmp
root=$(pwd); out=$root/gpurun_out/r05s18; mkdir -p $out
cd /tmp
for m in chain pair; do
rocprofv3 --kernel-trace --stats --output-format csv -d $out/$m -- python3 $root/scratch/r05/chain_only.py $m > $out/$m.log 2>&1
python3 - $out/$m <<'PY'
import csv, glob, re, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:4]:
    print("  %-90s calls %4s avg %8.2f us" % (re.sub(r"\(.*", "", r["Name"])[:90], r["Calls"], float(r["AverageNs"]) / 1e3))
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
rows = [r for r in rows if "spmm_" in r["Kernel_Name"]][-12:]
t0 = int(rows[0]["Start_Timestamp"])
for r in rows:
    print("   +%8.2f us  dur %7.2f  %s" % ((int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3, re.sub(r"\(.*", "", r["Kernel_Name"])[:60]))
PY
done
