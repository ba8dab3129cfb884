#!/bin/bash
export TMPDIR=/tmp
root=$(pwd); out=$root/gpurun_out/r05/asm_g5; mkdir -p $out
for m in AvI EvI IvE; do
  (cd /tmp && rocprofv3 --kernel-trace --output-format csv -d $out/$m -- python3 $root/scratch/time_assembly.py g5 $m > $out/$m.log 2>&1)
  python3 - <<PY
import csv, glob
f = glob.glob("$out/$m/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
# the last build = the kernels after the last k_fa_init
idx = [i for i, r in enumerate(rows) if "k_fa_init" in r["Kernel_Name"]]
last = rows[idx[-1]:]
t0 = int(last[0]["Start_Timestamp"])
busy = 0
print("== $m: last build, %d kernels" % len(last))
for r in last:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    busy += e - s
    print("  +%7.1f us  %6.1f us  %s" % ((s - t0) / 1e3, (e - s) / 1e3, r["Kernel_Name"].split("(")[0][:70]))
print("  span %.1f us, GPU busy %.1f us" % ((int(last[-1]["End_Timestamp"]) - t0) / 1e3, busy / 1e3))
PY
done
