#!/bin/bash
# usage: scratch/pmc.sh <outdir-name> <counter> <program> <args...>   (run on the GPU box; one counter group per pass)
set -e
out=$1; shift
ctr=$1; shift
export TMPDIR=/tmp
root=$(pwd)
mkdir -p gpurun_out/$out
cd /tmp
rocprofv3 --pmc $ctr --output-format csv -d $root/gpurun_out/$out -- "$@" > $root/gpurun_out/$out/run.log 2>&1
cd $root
find gpurun_out/$out -name '*counter_collection*' | head -n 3
