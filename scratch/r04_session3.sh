#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/s3
PYTHONFAULTHANDLER=1 python -m pytest tests/test_distributed_gloo.py -k custom_transport -x -q > gpurun_out/s3/dbg.log 2>&1
grep -n "Fatal\|File \|Segmentation\|passed\|failed" gpurun_out/s3/dbg.log | head -30
python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "streamed_build or random_sorted_grids" > gpurun_out/s3/pytest_stream.log 2>&1
echo "pytest stream rc=$?"; tail -3 gpurun_out/s3/pytest_stream.log
for m in AvI IvE; do
  DIV=6 bash scratch/prof_asm.sh r04b_$m $(pwd)/scratch/time_assembly.py a1h $m > gpurun_out/s3/kern_$m.txt 2>&1 || exit 1
  grep "k_sa\|k_fa\|scan" gpurun_out/s3/kern_$m.txt | head -12; grep "a1h" gpurun_out/prof_asm_r04b_$m/run.log
done
