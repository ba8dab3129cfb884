#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/s46
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_distributed_gloo.py -x -q -m gpu -k "streamed_build or random_sorted_grids or sharded_assembly or config5_full_size" > gpurun_out/s46/pytest.log 2>&1
rc=$?; echo "pytest rc=$rc"; tail -4 gpurun_out/s46/pytest.log
[ $rc -eq 0 ] || exit 1
python scratch/time_assembly.py a1h AvI,IvA,EvI,IvE,AvX,XvE 2>&1 | grep -v amdgpu | tee gpurun_out/s46/times.txt
python scratch/time_assembly.py g1 AvI,IvA,EvI,IvE 2>&1 | grep -v amdgpu | tee -a gpurun_out/s46/times.txt
