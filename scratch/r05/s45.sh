#!/bin/bash
for m in AvI EvI IvA IvE XvE; do
python3 scratch/r05/build_trace.py g5 $m 2>&1 | grep "per build"
TUNE=assemble_stream=1 python3 scratch/r05/build_trace.py g5 $m 2>&1 | grep "per build" | sed 's/$/  [streamed build forced]/'
done
