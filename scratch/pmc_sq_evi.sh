#!/bin/bash
# SQ counters of the column sweep at 1 km EvI (one apply per launch): where the wave cycles go
# (WAIT_ANY = parked on s_waitcnt / barrier, WAIT_INST_ANY = issue stalls, ACTIVE_INST_ANY = issuing) -- rocprofv3 --pmc alone
set -e
export TMPDIR=/tmp
root=$(pwd); out=$root/gpurun_out/pmc_sq_evi; mkdir -p $out
cd /tmp
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVES --output-format csv -d $out/p1 -- python3 $root/scratch/evi_one.py g1 colsweep 1 8 > $out/p1.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM --output-format csv -d $out/p2 -- python3 $root/scratch/evi_one.py g1 colsweep 1 8 > $out/p2.log 2>&1
cd $root
python3 - <<'PY'
import csv, glob, re
for p in ("p1", "p2"):
    fs = glob.glob("gpurun_out/pmc_sq_evi/%s/**/*counter_collection.csv" % p, recursive=True)
    agg = {}
    for r in csv.DictReader(open(fs[0])):
        k = re.sub(r"\(.*", "", r["Kernel_Name"])
        if "sweep" not in k and "rowblock" not in k: continue
        agg.setdefault((k, r["Counter_Name"]), []).append(float(r["Counter_Value"]))
    for (k, c), v in sorted(agg.items()):
        print("%-44s %-22s n=%2d mean %.4e" % (k[:44], c, len(v), sum(v) / len(v)))
PY
