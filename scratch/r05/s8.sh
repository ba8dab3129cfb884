#!/bin/bash
mkdir -p gpurun_out/r05s8; o=gpurun_out/r05s8; : > $o/tl.txt
export ICEBIN_HIP_LIB=$PWD/icebin_amd/lib/libicebin_hip_tl.so
for a in "g5 EvI 16 grouptile_fields=16" "g1 EvI 64 grouptile_fields=16"; do
  timeout -k 10 200 python scratch/r05/timeline.py $a >> $o/tl.txt 2>&1
done
grep -v amdgpu.ids $o/tl.txt | grep "tile 0\|whole\|proc0"
