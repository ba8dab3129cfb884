#!/bin/bash
mkdir -p gpurun_out/r05s19; o=gpurun_out/r05s19
for v in "" _sl127; do
ICEBIN_HIP_LIB=$PWD/icebin_amd/lib/libicebin_hip$v.so timeout -k 10 300 python scratch/chain3.py 16 g5 2>&1 | grep "chain call" | sed "s/^/[$v]/"
done
