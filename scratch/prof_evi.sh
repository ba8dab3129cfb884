#!/bin/bash
# 1 km EvI, 64 fields: kernel-trace stats of the three kernels and FETCH_SIZE / WRITE_SIZE passes of the column sweep
# (rocprofv3 --pmc alone, separate passes).  Summary: scratch/pmc_evi_summary.py
set -e
export TMPDIR=/tmp
root=$(pwd)
out=$root/gpurun_out/prof_evi
mkdir -p $out
cd /tmp
for mode in rowblock rowdual colsweep; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $out/kt_$mode -- python3 $root/scratch/evi_one.py g1 $mode 16 4 > $out/kt_$mode.log 2>&1
done
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/pmc_fetch -- python3 $root/scratch/evi_one.py g1 colsweep 1 12 > $out/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/pmc_write -- python3 $root/scratch/evi_one.py g1 colsweep 1 12 > $out/pmc_write.log 2>&1
cd $root
python3 scratch/pmc_evi_summary.py $out > $out/summary.txt
cat $out/summary.txt
