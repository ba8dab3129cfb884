#!/bin/bash
# kernels of the coupler's IvE {identity dimI, dimE as EvI left it} under the kernel trace
export TMPDIR=/tmp
root=$(pwd); out=$root/gpurun_out/r05coupler; mkdir -p $out
cd /tmp
rocprofv3 --kernel-trace --output-format csv -d $out/t -- python3 $root/scratch/r05/coupler_trace.py ${1:-g5} > $out/t.log 2>&1
python3 - $out/t <<'PY'
import csv, glob, re, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "k_fa_count" in r["Kernel_Name"]]
for base in idx[-2:]:
    nxt = [i for i in idx if i > base]
    seg = rows[base:(nxt[0] if nxt else len(rows))]
    t0 = int(seg[0]["Start_Timestamp"])
    for r in seg:
        print("   +%8.2f us  dur %7.2f  grid %6s wg %5s  %s" % ((int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3, r.get("Grid_Size_X", ""), r.get("Workgroup_Size_X", ""), re.sub(r"\(.*", "", r["Kernel_Name"])[:70]))
    print()
PY
