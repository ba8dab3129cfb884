#!/bin/bash
# soak runs on the round's new paths: tiled row groups (both tile sizes), the chained scan inside the builds
out=gpurun_out/r05s31; mkdir -p $out
TUNE=rowgroup_form=1 timeout -k 10 300 python scratch/fuzz_applies.py 100 40 > $out/applies_tiles256.log 2>&1; echo "rc $?" >> $out/applies_tiles256.log
TUNE=rowgroup_form=1,grouptile_seg=128 timeout -k 10 300 python scratch/fuzz_applies.py 140 40 > $out/applies_tiles128.log 2>&1; echo "rc $?" >> $out/applies_tiles128.log
timeout -k 10 400 python scratch/fuzz_builds.py 200 40 > $out/builds.log 2>&1; echo "rc $?" >> $out/builds.log
tail -n 4 $out/applies_tiles256.log $out/applies_tiles128.log $out/builds.log
