#!/bin/bash
# round 4, GPU session 1: streamed-build parity on the small grids, timing of the Antarctic builds (both paths)
set -o pipefail
mkdir -p gpurun_out/s1
python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "streamed_build or random_sorted_grids" > gpurun_out/s1/pytest_stream.log 2>&1
echo "pytest stream rc=$?" | tee -a gpurun_out/s1/summary.txt
tail -5 gpurun_out/s1/pytest_stream.log
python scratch/time_assembly.py a1h AvI,IvA,EvI,IvE,AvX,XvE > gpurun_out/s1/asm_stream.txt 2>&1
echo "asm rc=$?" | tee -a gpurun_out/s1/summary.txt
cat gpurun_out/s1/asm_stream.txt
