#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/s2
for m in AvI IvE; do
  DIV=6 bash scratch/prof_asm.sh r04_$m $(pwd)/scratch/time_assembly.py a1h $m > gpurun_out/s2/kern_$m.txt 2>&1 || exit 1
  cat gpurun_out/s2/kern_$m.txt | head -24
done
python -m pytest tests -x -q -m gpu > gpurun_out/s2/pytest_gpu.log 2>&1
echo "pytest rc=$?"
tail -8 gpurun_out/s2/pytest_gpu.log
