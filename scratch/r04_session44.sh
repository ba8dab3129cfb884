#!/bin/bash
mkdir -p gpurun_out/s44
timeout -k 10 500 python scratch/fuzz_applies.py 0 400 2>&1 | grep -v amdgpu > gpurun_out/s44/fuzz_default.log; echo "default rc=$?"; tail -2 gpurun_out/s44/fuzz_default.log
TUNE=shortrow_xt=1,shortrow_group=4 timeout -k 10 500 python scratch/fuzz_applies.py 1000 400 2>&1 | grep -v amdgpu > gpurun_out/s44/fuzz_xt.log; echo "xt rc=$?"; tail -2 gpurun_out/s44/fuzz_xt.log
grep -c MISMATCH gpurun_out/s44/fuzz_default.log gpurun_out/s44/fuzz_xt.log
