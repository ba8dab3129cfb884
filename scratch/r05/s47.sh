#!/bin/bash
python3 scratch/time_assembly.py a1h,g1 IvE,XvE 2>&1 | grep -v "amdgpu\|regrid_matrices"
export ICEBIN_HIP_LIB=$PWD/icebin_amd/lib/libicebin_hip_keep4.so
python3 scratch/time_assembly.py a1h,g1 IvE,XvE 2>&1 | grep -v "amdgpu\|regrid_matrices" | sed 's/$/  [keep 4]/'
