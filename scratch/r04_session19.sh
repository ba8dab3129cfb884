#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/s19
python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "streamed_build or random_sorted_grids or fast_and_general or matrix_batch" > gpurun_out/s19/pytest_stream.log 2>&1
echo "pytest stream rc=$?"; tail -12 gpurun_out/s19/pytest_stream.log
python scratch/coupler_step.py g1,a1h 2>&1 | grep -v amdgpu > gpurun_out/s19/coupler.txt; cat gpurun_out/s19/coupler.txt
TUNE=assemble_stream=0 python scratch/coupler_step.py a1h 2>&1 | grep -v amdgpu > gpurun_out/s19/coupler_old.txt; cat gpurun_out/s19/coupler_old.txt
