#!/bin/bash
# SQ / cache counters of the Antarctic IvE and IvA applies at 128 fields (rocprofv3 --pmc alone)
export TMPDIR=/tmp
root=$(pwd); out=$root/gpurun_out/s38; mkdir -p $out
cd /tmp
for m in IvE IvA; do
  rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_VALU --output-format csv -d $out/${m}_sq -- python3 $root/scratch/kernel_choice.py a1h $m 128 auto > $out/${m}_sq.log 2>&1
  rocprofv3 --pmc TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_EA_RDREQ_sum TCC_EA_WRREQ_sum TCC_REQ_sum --output-format csv -d $out/${m}_tc -- python3 $root/scratch/kernel_choice.py a1h $m 128 auto > $out/${m}_tc.log 2>&1
done
cd $root
python3 - <<'PY'
import csv, glob, re
for m in ("IvE", "IvA"):
    for t in ("sq", "tc"):
        fs = glob.glob("gpurun_out/s38/%s_%s/**/*counter_collection.csv" % (m, t), recursive=True)
        if not fs: print(m, t, "no csv:", open("gpurun_out/s38/%s_%s.log" % (m, t)).read()[-600:]); continue
        agg = {}
        for r in csv.DictReader(open(fs[0])):
            k = re.sub(r"\(.*", "", r["Kernel_Name"])
            if "shortrow" not in k: continue
            agg.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
        print(m, t, "  ".join("%s=%.3e" % (c, sum(v) / len(v)) for c, v in sorted(agg.items())))
PY
