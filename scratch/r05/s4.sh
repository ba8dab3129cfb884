#!/bin/bash
mkdir -p gpurun_out/r05s4; o=gpurun_out/r05s4; : > $o/tl.txt
export ICEBIN_HIP_LIB=$PWD/icebin_amd/lib/libicebin_hip_tl.so
for a in "g5 EvI 16 grouptile_fields=16" "g5 EvI 64 grouptile_fields=16" "g1 EvI 64 grouptile_fields=16" "g1 EvI 64 grouptile_fields=32"; do
  timeout -k 10 200 python scratch/r05/timeline.py $a >> $o/tl.txt 2>&1
done
cat $o/tl.txt
