#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/s16
python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "streamed_build or random_sorted_grids or full_size_assembly" > gpurun_out/s16/pytest_stream.log 2>&1
echo "pytest stream rc=$?"; tail -5 gpurun_out/s16/pytest_stream.log
python scratch/time_assembly.py a1h AvI,IvA,EvI,IvE,AvX,XvE > gpurun_out/s16/asm_stream.txt 2>&1; cat gpurun_out/s16/asm_stream.txt
python scratch/time_assembly.py g1,g1h AvI,IvA,EvI,IvE > gpurun_out/s16/asm_g1.txt 2>&1; cat gpurun_out/s16/asm_g1.txt
for m in AvI EvI; do
  DIV=6 bash scratch/prof_asm.sh r04f_$m $(pwd)/scratch/time_assembly.py a1h $m > gpurun_out/s16/kern_$m.txt 2>&1 || exit 1
  grep "k_sa\|k_fa" gpurun_out/s16/kern_$m.txt | head -7
done
