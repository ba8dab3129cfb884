#!/bin/bash
# longer soak: 300 seeds per form, 150 seeds of builds
out=gpurun_out/r05s32; mkdir -p $out
TUNE=rowgroup_form=1 timeout -k 10 500 python scratch/fuzz_applies.py 1000 300 > $out/applies_tiles256.log 2>&1; echo "rc $?" >> $out/applies_tiles256.log
TUNE=rowgroup_form=1,grouptile_seg=128 timeout -k 10 500 python scratch/fuzz_applies.py 2000 300 > $out/applies_tiles128.log 2>&1; echo "rc $?" >> $out/applies_tiles128.log
timeout -k 10 500 python scratch/fuzz_applies.py 3000 300 > $out/applies_auto.log 2>&1; echo "rc $?" >> $out/applies_auto.log
timeout -k 10 600 python scratch/fuzz_builds.py 1000 150 > $out/builds.log 2>&1; echo "rc $?" >> $out/builds.log
tail -n 4 $out/applies_tiles256.log $out/applies_tiles128.log $out/applies_auto.log $out/builds.log
grep -h "MISMATCH\|refused\|!=" $out/*.log | sort | uniq -c | sort -rn | head -20
true
