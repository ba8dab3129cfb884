#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/s32
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_distributed_gloo.py -x -q -m gpu -k "streamed_build or random_sorted_grids or sharded_assembly or config5_full_size or coupler" > gpurun_out/s32/pytest.log 2>&1
rc=$?; echo "pytest rc=$rc"; tail -5 gpurun_out/s32/pytest.log
[ $rc -eq 0 ] || exit 1
python scratch/time_assembly.py a1h AvI,IvA,EvI,IvE,AvX,XvE 2>&1 | grep -v amdgpu > gpurun_out/s32/times.txt
python scratch/time_assembly.py g1 AvI,IvA,EvI,IvE 2>&1 | grep -v amdgpu >> gpurun_out/s32/times.txt
cat gpurun_out/s32/times.txt
python scratch/coupler_step.py g1,a1h 2>&1 | grep -v amdgpu > gpurun_out/s32/coupler.txt; cat gpurun_out/s32/coupler.txt
