#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/s4
python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "streamed_build or random_sorted_grids" > gpurun_out/s4/pytest_stream.log 2>&1
echo "pytest stream rc=$?"; tail -3 gpurun_out/s4/pytest_stream.log
for m in AvI IvE; do
  DIV=6 bash scratch/prof_asm.sh r04c_$m $(pwd)/scratch/time_assembly.py a1h $m > gpurun_out/s4/kern_$m.txt 2>&1 || exit 1
  grep "k_sa\|k_fa" gpurun_out/s4/kern_$m.txt | head -8; grep "a1h" gpurun_out/prof_asm_r04c_$m/run.log
done
python scratch/time_assembly.py a1h AvI,IvA,EvI,IvE,AvX,XvE > gpurun_out/s4/asm_stream.txt 2>&1; cat gpurun_out/s4/asm_stream.txt
python scratch/time_assembly.py g1 AvI,IvA,EvI,IvE > gpurun_out/s4/asm_g1.txt 2>&1; cat gpurun_out/s4/asm_g1.txt
python -m pytest tests -x -q -m gpu > gpurun_out/s4/pytest_gpu.log 2>&1
echo "pytest rc=$?"
tail -5 gpurun_out/s4/pytest_gpu.log
