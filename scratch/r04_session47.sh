#!/bin/bash
mkdir -p gpurun_out/s47; o=gpurun_out/s47/g5.txt; : > $o
for t in "" shortrow_group=4 "shortrow_group=4,shortrow_xt=1" "shortrow_group=4,shortrow_xt=1,shortrow_fper=8"; do
  for m in IvA IvE; do TUNE=$t python scratch/kernel_choice.py g5 $m 16,64 auto 2>&1 | grep nf= | sed "s/^/[$t] /" >> $o; done
done
cat $o
