#!/bin/bash
# usage: scratch/prof.sh <outdir-name> <python args...>   (run on the GPU box)
set -e
out=$1; shift
export TMPDIR=/tmp
root=$(pwd)
mkdir -p gpurun_out/$out
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $root/gpurun_out/$out -- python3 "$@" > $root/gpurun_out/$out/run.log 2>&1
cd $root
find gpurun_out/$out -name '*kernel_stats*' | head
