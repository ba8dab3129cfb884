#!/bin/bash
# kernel-level profile of the assembly: usage scratch/prof_asm.sh <tag> <script args...>
export TMPDIR=/tmp
root=$(pwd); tag=$1; shift
out=$root/gpurun_out/prof_asm_$tag
mkdir -p $out
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $out -- python3 "$@" > $out/run.log 2>&1
cd $root
f=$(find $out -name '*kernel_stats.csv' | head -1)
python3 scratch/kstats.py $f ${DIV:-1} 40
