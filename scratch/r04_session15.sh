#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/s15
python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "streamed_build or random_sorted_grids or full_size_assembly" > gpurun_out/s15/pytest_stream.log 2>&1
echo "pytest stream rc=$?"; tail -3 gpurun_out/s15/pytest_stream.log
timeout -k 10 600 python -m pytest tests/test_distributed_gloo.py -x -q -m gpu -k sharded_assembly > gpurun_out/s15/pytest_dist.log 2>&1
echo "pytest dist rc=$?"; tail -3 gpurun_out/s15/pytest_dist.log
python scratch/time_assembly.py a1h AvI,IvA,EvI,IvE,AvX,XvE > gpurun_out/s15/asm_stream.txt 2>&1; cat gpurun_out/s15/asm_stream.txt
python scratch/time_assembly.py g1 AvI,IvA,EvI,IvE > gpurun_out/s15/asm_g1.txt 2>&1; cat gpurun_out/s15/asm_g1.txt
