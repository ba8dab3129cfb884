#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/s17
python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "streamed_build or random_sorted_grids or full_size_assembly" > gpurun_out/s17/pytest_stream.log 2>&1
echo "pytest stream rc=$?"; tail -3 gpurun_out/s17/pytest_stream.log
for c in 1 0 1 0; do echo rows4=$c; TUNE=assemble_stream_rows4=$c python scratch/time_assembly.py a1h AvI,IvA,EvI,IvE,XvE 2>&1 | grep a1h; done
python scratch/time_assembly.py g1,g1h AvI,IvA,EvI,IvE > gpurun_out/s17/asm_g1.txt 2>&1; cat gpurun_out/s17/asm_g1.txt
for m in AvI EvI; do
  DIV=6 bash scratch/prof_asm.sh r04g_$m $(pwd)/scratch/time_assembly.py a1h $m > gpurun_out/s17/kern_$m.txt 2>&1 || exit 1
  grep "k_sa_rows\|k_sa_emit" gpurun_out/s17/kern_$m.txt | head -3
done
