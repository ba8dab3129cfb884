#!/bin/bash
for c in g5 g20; do
for m in AvI EvI IvE XvE; do
for sh in 1 2 3; do
TUNE=assemble_range_shape=$sh python3 scratch/r05/build_trace.py $c $m 2>&1 | grep "per build" | sed "s/$/ shape $sh/"
done; done; done
