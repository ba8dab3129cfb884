#!/bin/bash
# SQ / traffic counters of one EvI apply per launch: pmc.sh tag cfg nf "TUNE"   (rocprofv3 --pmc alone, separate passes)
tag=$1; cfg=$2; nf=$3; export TUNE=$4
export TMPDIR=/tmp
root=$(pwd); out=$root/gpurun_out/r05pmc/$tag; mkdir -p $out
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $out/kt -- python3 $root/scratch/r05/one.py $cfg EvI $nf rowgroup 12 > $out/kt.log 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVES --output-format csv -d $out/p1 -- python3 $root/scratch/r05/one.py $cfg EvI $nf rowgroup 8 > $out/p1.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM --output-format csv -d $out/p2 -- python3 $root/scratch/r05/one.py $cfg EvI $nf rowgroup 8 > $out/p2.log 2>&1
rocprofv3 --pmc SQ_BUSY_CYCLES SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INSTS_BRANCH SQ_WAIT_INST_VMEM GRBM_GUI_ACTIVE --output-format csv -d $out/p3 -- python3 $root/scratch/r05/one.py $cfg EvI $nf rowgroup 8 > $out/p3.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/pf -- python3 $root/scratch/r05/one.py $cfg EvI $nf rowgroup 8 > $out/pf.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/pw -- python3 $root/scratch/r05/one.py $cfg EvI $nf rowgroup 8 > $out/pw.log 2>&1
cd $root
python3 - $out <<'PY' > $out/summary.txt
import csv, glob, re, sys, os
out = sys.argv[1]
print(open(out + "/kt.log").read().strip().splitlines()[-1])
fs = glob.glob(out + "/kt/**/*kernel_stats.csv", recursive=True)
for r in list(csv.DictReader(open(fs[0])))[:4]:
    print("  %-70s calls %4s avg %10.1f us" % (re.sub(r"\(.*", "", r["Name"])[:70], r["Calls"], float(r["AverageNs"]) / 1e3))
for p in ("p1", "p2", "p3", "pf", "pw"):
    fs = glob.glob(out + "/%s/**/*counter_collection.csv" % p, recursive=True)
    if not fs: print(p, "no csv"); continue
    agg = {}
    for r in csv.DictReader(open(fs[0])):
        k = re.sub(r"\(.*", "", r["Kernel_Name"])
        if "spmm_" not in k and "elementwise" not in k: continue
        agg.setdefault((k, r["Counter_Name"]), []).append(float(r["Counter_Value"]))
    for (k, c), v in sorted(agg.items()):
        print("%-50s %-22s n=%2d mean %.4e" % (k[:50], c, len(v), sum(v) / len(v)))
PY
cat $out/summary.txt
