#!/bin/bash
# soak runs: applies (tiled and automatic row groups), builds (every variant of the plan-based build), sharded builds at 2-4 ranks -- random grids against the oracle
out=gpurun_out/r05soak; mkdir -p $out
TUNE=rowgroup_form=1 timeout -k 10 300 python scratch/fuzz_applies.py 5000 150 > $out/applies_tiles.log 2>&1; echo "rc $?" >> $out/applies_tiles.log
timeout -k 10 300 python scratch/fuzz_applies.py 6000 150 > $out/applies_auto.log 2>&1; echo "rc $?" >> $out/applies_auto.log
timeout -k 10 600 python scratch/fuzz_builds.py 3000 300 > $out/builds.log 2>&1; echo "rc $?" >> $out/builds.log
timeout -k 10 900 python scratch/fuzz_sharded.py 1000 150 15 > $out/sharded.log 2>&1; echo "rc $?" >> $out/sharded.log
tail -n 3 $out/applies_tiles.log $out/applies_auto.log $out/builds.log; grep "^batches\|^rc" $out/sharded.log
