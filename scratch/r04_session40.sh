#!/bin/bash
mkdir -p gpurun_out/s40; o=gpurun_out/s40/xcd.txt; : > $o
for t in shortrow_xcd=1 shortrow_xcd=0 "shortrow_xcd=0,shortrow_fper=128" "shortrow_xcd=0,shortrow_fper=64" "shortrow_xcd=0,shortrow_fper=128,shortrow_group=8"; do
  for m in IvE IvA; do
    TUNE=$t python scratch/kernel_choice.py a1h $m 16,128 auto 2>&1 | grep nf= | sed "s/^/$t /" >> $o
  done
done
TUNE=shortrow_xcd=0 python scratch/kernel_choice.py g1 IvE 16,64 auto 2>&1 | grep nf= | sed "s/^/xcd=0 /" >> $o
TUNE=shortrow_xcd=1 python scratch/kernel_choice.py g1 IvE 16,64 auto 2>&1 | grep nf= | sed "s/^/xcd=1 /" >> $o
cat $o
