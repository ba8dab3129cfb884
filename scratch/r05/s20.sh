#!/bin/bash
mkdir -p gpurun_out/r05s20; o=gpurun_out/r05s20; : > $o/a1h.txt
for t in "" "rowblock_xcd_mode=0" "grouptile_seg=256"; do
  TUNE=$t timeout -k 10 400 python scratch/kernel_choice.py a1h EvI 16,128 rowgroup 2>&1 | grep nf= | sed "s/^/[$t] /" >> $o/a1h.txt
done
cat $o/a1h.txt
