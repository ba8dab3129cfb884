#!/bin/bash
# rowgroup: all field chunks of a group on ONE XCD (rowblock_xcd_mode=0) against one chunk per XCD (mode 1, default)
set -o pipefail
mkdir -p gpurun_out/s20
o=gpurun_out/s20/xcd.txt; : > $o
for t in "" "rowblock_xcd_mode=0"; do
  TUNE=$t python scratch/one_matrix.py g1 EvI 2>&1 | grep -v amdgpu >> $o
  NF=128 TUNE=$t python scratch/one_matrix.py g1 EvI 2>&1 | grep -v amdgpu >> $o
  TUNE=$t python scratch/one_matrix.py g5 EvI 2>&1 | grep -v amdgpu >> $o
  NF=16 TUNE=$t python scratch/one_matrix.py g5 EvI 2>&1 | grep -v amdgpu >> $o
  TUNE=$t python scratch/one_matrix.py g1 AvI 2>&1 | grep -v amdgpu >> $o
done
cat $o
