#!/bin/bash
# Antarctic EvI: the LDS-atomic row groups against the tiled form
mkdir -p gpurun_out/r05s9; o=gpurun_out/r05s9; : > $o/a1h.txt
for t in rowgroup_form=0 rowgroup_form=1 "rowgroup_form=1,grouptile_fields=16"; do
  TUNE=$t timeout -k 10 400 python scratch/kernel_choice.py a1h EvI 16,128 rowgroup 2>&1 | grep nf= | sed "s/^/[$t] /" >> $o/a1h.txt
done
cat $o/a1h.txt
