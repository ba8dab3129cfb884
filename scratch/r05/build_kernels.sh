#!/bin/bash
export TMPDIR=/tmp
root=$(pwd); out=$root/gpurun_out/r05build; mkdir -p $out
cd /tmp
for m in EvI IvE; do
rocprofv3 --kernel-trace --output-format csv -d $out/$m -- python3 $root/scratch/r05/build_trace.py g5 $m > $out/$m.log 2>&1
python3 - $out/$m > $out/$m.trace.txt <<'PY'
import csv, glob, re, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "k_fa_count" in r["Kernel_Name"]]
for base in idx[-3:-1]:
    seg = rows[base:idx[idx.index(base) + 1]]
    t0 = int(seg[0]["Start_Timestamp"])
    for r in seg:
        print("   +%8.2f us  dur %7.2f  grid %6s wg %5s  %s" % ((int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3, r.get("Grid_Size_X", r.get("Grid_Size", "")), r.get("Workgroup_Size_X", r.get("Workgroup_Size", "")), re.sub(r"\(.*", "", r["Kernel_Name"])[:70]))
    print()
PY
done
cat $out/*.trace.txt
