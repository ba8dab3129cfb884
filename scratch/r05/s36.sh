#!/bin/bash
out=gpurun_out/r05s36; mkdir -p $out
export ICEBIN_HIP_LIB=$PWD/icebin_amd/lib/libicebin_hip_fatl.so
for m in AvI EvI IvE; do timeout -k 10 200 python scratch/r05/range_timeline.py g5 $m 2>&1 | grep -v amdgpu.ids; done > $out/tl.txt
cat $out/tl.txt
