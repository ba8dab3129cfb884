#!/bin/bash
root=$(pwd); out=$root/gpurun_out/r04b; mkdir -p $out
export TMPDIR=/tmp
(cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $out/empty_kt -- $root/scratch/empty_bench > $out/empty_kt.log 2>&1)
python3 scratch/rocsum.py $out/empty_kt "k_empty" > $out/empty_launch_rocprof.txt; cat $out/empty_launch_rocprof.txt | head -20; grep EMPTY $out/empty_kt.log
./scratch/empty_bench > $out/empty_noprof.log 2>&1; cat $out/empty_noprof.log
