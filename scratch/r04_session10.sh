#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/s10
PYTHONFAULTHANDLER=1 timeout -k 10 900 python -m pytest tests/test_distributed_gloo.py -x -q -m gpu > gpurun_out/s10/pytest_dist.log 2>&1
echo "pytest dist rc=$?"; tail -25 gpurun_out/s10/pytest_dist.log
python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "streamed_build or random_sorted_grids or full_size_assembly" > gpurun_out/s10/pytest_stream.log 2>&1
echo "pytest stream rc=$?"; tail -3 gpurun_out/s10/pytest_stream.log
python scratch/time_assembly.py a1h AvI,IvA,EvI,IvE > gpurun_out/s10/asm_stream.txt 2>&1; cat gpurun_out/s10/asm_stream.txt
