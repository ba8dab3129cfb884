#!/bin/bash
mkdir -p gpurun_out/s45; o=gpurun_out/s45/g1.txt; : > $o
for t in "" shortrow_group=4 "shortrow_group=4,shortrow_fper=32" "shortrow_group=4,shortrow_fper=64"; do
  for m in IvA IvE; do TUNE=$t python scratch/kernel_choice.py g1 $m 16,64 auto 2>&1 | grep nf= | sed "s/^/[$t] /" >> $o; done
done
for t in "" shortrow_group=4; do
  for m in IvA IvE; do TUNE=$t python scratch/kernel_choice.py g1h $m 64 auto 2>&1 | grep nf= | sed "s/^/[$t] /" >> $o; done
done
cat $o
