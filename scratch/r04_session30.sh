#!/bin/bash
# SQ counters of the streamed build's kernels (a1h AvI, IvE): instruction mix and how busy the vector ALUs are (rocprofv3 --pmc alone)
set -o pipefail
export TMPDIR=/tmp
root=$(pwd); out=$root/gpurun_out/s30; mkdir -p $out
cd /tmp
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SMEM --output-format csv -d $out/p1 -- python3 $root/scratch/time_assembly.py a1h AvI,IvE > $out/p1.log 2>&1
rocprofv3 --pmc SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d $out/p2 -- python3 $root/scratch/time_assembly.py a1h AvI,IvE > $out/p2.log 2>&1
cd $root
python3 - <<'PY'
import csv, glob, re
for p in ("p1", "p2"):
    fs = glob.glob("gpurun_out/s30/%s/**/*counter_collection.csv" % p, recursive=True)
    if not fs: print(p, "no csv"); print(open("gpurun_out/s30/%s.log" % p).read()[-1500:]); continue
    agg = {}
    for r in csv.DictReader(open(fs[0])):
        k = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void ibh::", "")
        if not k.startswith("k_sa_"): continue
        agg.setdefault((k, r["Counter_Name"]), []).append(float(r["Counter_Value"]))
    for (k, c), v in sorted(agg.items()):
        print("%-50s %-22s n=%2d mean %.4e" % (k[:50], c, len(v), sum(v) / len(v)))
PY
