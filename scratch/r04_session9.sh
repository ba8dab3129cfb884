#!/bin/bash
# one apply() per launch under rocprofv3, PLAIN launches (no HIP events attached) vs event-wrapped launches
root=$(pwd); out=$root/gpurun_out/r04b; mkdir -p $out
export TMPDIR=/tmp
for mode in plain events; do
  extra=""; [ $mode = plain ] && extra="--no-kernel-events"
  (cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $out/kt_depth1_$mode -- python3 $root/bench.py --queue-depth 1 --steps 320 --warmup 32 --repeats 5 --no-cpu-baseline --no-extras $extra > $out/kt_depth1_$mode.log 2>&1)
  python3 scratch/rocsum.py $out/kt_depth1_$mode spmm_ > $out/kt_depth1_$mode.summary.txt; head -6 $out/kt_depth1_$mode.summary.txt
done
for mode in plain events; do
  extra=""; [ $mode = plain ] && extra="--no-kernel-events"
  (cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $out/kt_default_$mode -- python3 $root/bench.py --no-cpu-baseline --no-extras --repeats 5 $extra > $out/kt_default_$mode.log 2>&1)
  python3 scratch/rocsum.py $out/kt_default_$mode spmm_ > $out/kt_default_$mode.summary.txt; head -6 $out/kt_default_$mode.summary.txt
done
find $out -name "*.csv" -size +4M -delete
