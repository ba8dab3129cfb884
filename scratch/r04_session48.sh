#!/bin/bash
mkdir -p gpurun_out/s48; o=gpurun_out/s48/rg.txt; : > $o
for t in "" rowgroup_waves=4 rowgroup_waves=8 rowgroup_tw=32 rowgroup_tw=64 "rowgroup_waves=8,rowgroup_tw=32" "rowgroup_waves=4,rowgroup_tw=64" rowgroup_unroll=4 rowgroup_unroll=12; do
  TUNE=$t python scratch/kernel_choice.py a1h EvI 16,128 rowgroup 2>&1 | grep nf= | sed "s/^/[$t] /" >> $o
done
cat $o
