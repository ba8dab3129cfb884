#!/bin/bash
# PMC traffic of the Antarctic IvE / IvA applies at 128 fields (rocprofv3 --pmc alone, separate passes)
export TMPDIR=/tmp
root=$(pwd); out=$root/gpurun_out/s37; mkdir -p $out
cd /tmp
for m in IvE IvA; do
  for c in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --pmc $c --output-format csv -d $out/${m}_$c -- python3 $root/scratch/kernel_choice.py a1h $m 128 auto > $out/${m}_$c.log 2>&1
  done
done
cd $root
python3 - <<'PY'
import csv, glob, re
for m in ("IvE", "IvA"):
    for c in ("FETCH_SIZE", "WRITE_SIZE"):
        fs = glob.glob("gpurun_out/s37/%s_%s/**/*counter_collection.csv" % (m, c), recursive=True)
        agg = {}
        for r in csv.DictReader(open(fs[0])):
            k = re.sub(r"\(.*", "", r["Kernel_Name"])
            if "shortrow" not in k and "transpose" not in k: continue
            agg.setdefault(k[:60], []).append(float(r["Counter_Value"]))
        for k, v in agg.items():
            mb = sum(v) / len(v) * 1024 / 1e6 * (2.0 if c == "FETCH_SIZE" else 1.0)       # (KB units; FETCH_SIZE x 2 on gfx950, as scratch/rocsum.py)
            print(m, c, k, "n=%d" % len(v), "mean %.1f MB per launch" % mb)
PY
grep nf= $out/IvE_FETCH_SIZE.log | head -2
