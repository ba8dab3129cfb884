#!/bin/bash
export TMPDIR=/tmp
root=$(pwd); out=$root/gpurun_out/r05sq; mkdir -p $out
cd /tmp
for m in IvE IvA; do
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD --output-format csv -d $out/p1$m -- python3 $root/scratch/kernel_choice.py a1h $m 128 auto > $out/p1$m.log 2>&1
rocprofv3 --pmc SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS SQ_BUSY_CYCLES SQC_DCACHE_REQ SQC_DCACHE_MISSES --output-format csv -d $out/p2$m -- python3 $root/scratch/kernel_choice.py a1h $m 128 auto > $out/p2$m.log 2>&1
rocprofv3 --pmc TCP_PENDING_STALL_CYCLES_sum TA_BUSY_avr TCC_EA0_WRREQ_STALL_sum TCC_EA0_WRREQ_sum TCC_HIT_sum TCC_MISS_sum --output-format csv -d $out/p3$m -- python3 $root/scratch/kernel_choice.py a1h $m 128 auto > $out/p3$m.log 2>&1
done
cd $root
python3 - $out <<'PY'
import csv, glob, re, sys
out = sys.argv[1]
for m in ("IvE", "IvA"):
  for p in ("p1", "p2", "p3"):
    fs = glob.glob(out + "/%s%s/**/*counter_collection.csv" % (p, m), recursive=True)
    if not fs: print(p, m, "no csv"); continue
    agg = {}
    for r in csv.DictReader(open(fs[0])):
        k = re.sub(r"\(.*", "", r["Kernel_Name"])
        if "shortrow" not in k: continue
        agg.setdefault((k, r["Counter_Name"]), []).append(float(r["Counter_Value"]))
    for (k, c), v in sorted(agg.items()):
        print("%s %-44s %-28s n=%2d mean %.4e" % (m, k[-44:], c, len(v), sum(v) / len(v)))
PY
