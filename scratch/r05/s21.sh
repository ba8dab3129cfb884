#!/bin/bash
export TMPDIR=/tmp
root=$(pwd); out=$root/gpurun_out/r05s21; mkdir -p $out
cd /tmp
export TUNE=shortrow_fieldlane=1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/pf -- python3 $root/scratch/kernel_choice.py a1h IvE 128 auto > $out/pf.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/pw -- python3 $root/scratch/kernel_choice.py a1h IvE 128 auto > $out/pw.log 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD --output-format csv -d $out/p1 -- python3 $root/scratch/kernel_choice.py a1h IvE 128 auto > $out/p1.log 2>&1
rocprofv3 --pmc SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_BUSY_CYCLES TCP_PENDING_STALL_CYCLES_sum --output-format csv -d $out/p2 -- python3 $root/scratch/kernel_choice.py a1h IvE 128 auto > $out/p2.log 2>&1
cd $root
python3 - $out <<'PY'
import csv, glob, re, sys
out = sys.argv[1]
for p in ("pf", "pw", "p1", "p2"):
    fs = glob.glob(out + "/%s/**/*counter_collection.csv" % p, recursive=True)
    if not fs: print(p, "no csv"); continue
    agg = {}
    for r in csv.DictReader(open(fs[0])):
        k = re.sub(r"\(.*", "", r["Kernel_Name"])
        if "fieldlane" not in k and "shortrow" not in k and "transpose" not in k: continue
        agg.setdefault((k, r["Counter_Name"]), []).append(float(r["Counter_Value"]))
    for (k, c), v in sorted(agg.items()):
        print("%-40s %-26s n=%2d mean %.4e" % (k[-40:], c, len(v), sum(v) / len(v)))
PY
