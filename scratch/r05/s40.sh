#!/bin/bash
for c in g5; do
for m in AvI EvI IvE IvA XvE; do
python3 scratch/r05/build_trace.py $c $m 2>&1 | grep "per build"
done; done
python3 scratch/time_assembly.py a1h AvI,EvI,IvA,IvE 2>&1 | grep -v amdgpu | tail -6
