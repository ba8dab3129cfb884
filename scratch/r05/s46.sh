#!/bin/bash
for m in IvE XvE EvI; do python3 scratch/r05/build_trace.py g5 $m 2>&1 | grep "per build"; done
python3 scratch/time_assembly.py a1h IvE,IvA 2>&1 | grep -v "amdgpu\|regrid_matrices"
export ICEBIN_HIP_LIB=$PWD/icebin_amd/lib/libicebin_hip_fatl.so
for m in IvE; do timeout -k 10 200 python scratch/r05/pelem_timeline.py g5 $m 2>&1 | grep -v amdgpu.ids; done
