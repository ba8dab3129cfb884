#!/bin/bash
# Round-2 profile set for the headline bench (run on the GPU box from the repo root):
#   kernel-trace stats of the default bench and of the driver's 20-step command, then separate
#   FETCH_SIZE / WRITE_SIZE PMC passes (rocprofv3 --pmc alone), summarised by scratch/pmc_summary.py.
set -e
export TMPDIR=/tmp
root=$(pwd)
out=$root/gpurun_out/prof_r02
mkdir -p $out
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $out/kt_default -- python3 $root/bench.py --no-cpu-baseline > $out/kt_default.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $out/kt_driver20 -- python3 $root/bench.py --steps 20 --warmup 5 --no-cpu-baseline > $out/kt_driver20.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/pmc_fetch -- python3 $root/bench.py --steps 64 --warmup 32 --no-cpu-baseline > $out/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/pmc_write -- python3 $root/bench.py --steps 64 --warmup 32 --no-cpu-baseline > $out/pmc_write.log 2>&1
cd $root
python3 scratch/pmc_summary.py $out > $out/summary.txt
cat $out/summary.txt
