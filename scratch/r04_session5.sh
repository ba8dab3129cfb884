#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/s5
python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "streamed_build or random_sorted_grids or full_size_assembly" > gpurun_out/s5/pytest_stream.log 2>&1
echo "pytest stream rc=$?"; tail -3 gpurun_out/s5/pytest_stream.log
python scratch/time_assembly.py a1h AvI,IvA,EvI,IvE,AvX,XvE > gpurun_out/s5/asm_stream.txt 2>&1; cat gpurun_out/s5/asm_stream.txt
python scratch/time_assembly.py g1 AvI,IvA,EvI,IvE > gpurun_out/s5/asm_g1.txt 2>&1; cat gpurun_out/s5/asm_g1.txt
for m in EvI; do
  DIV=6 bash scratch/prof_asm.sh r04d_$m $(pwd)/scratch/time_assembly.py a1h $m > gpurun_out/s5/kern_$m.txt 2>&1 || exit 1
  grep "k_sa\|k_fa" gpurun_out/s5/kern_$m.txt | head -8
  DIV=6 bash scratch/prof_asm.sh r04d_g1_$m $(pwd)/scratch/time_assembly.py g1 $m > gpurun_out/s5/kern_g1_$m.txt 2>&1 || exit 1
  grep "k_sa\|k_fa" gpurun_out/s5/kern_g1_$m.txt | head -8
done
