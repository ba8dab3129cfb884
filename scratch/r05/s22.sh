#!/bin/bash
mkdir -p gpurun_out/r05s22; o=gpurun_out/r05s22; : > $o/a1h.txt
for v in "" _b8 _u32 _w6; do
  ICEBIN_HIP_LIB=$PWD/icebin_amd/lib/libicebin_hip$v.so timeout -k 10 400 python scratch/kernel_choice.py a1h EvI 16,128 rowgroup 2>&1 | grep nf= | sed "s/^/[$v] /" >> $o/a1h.txt
done
cat $o/a1h.txt
