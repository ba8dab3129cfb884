#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/s22
python -m pytest tests/test_gpu_parity.py tests/test_distributed_gloo.py -x -q -m gpu -k "streamed_build or random_sorted_grids or sharded_assembly or config5_full_size" > gpurun_out/s22/pytest.log 2>&1
echo "pytest rc=$?"; tail -5 gpurun_out/s22/pytest.log
python scratch/time_assembly.py a1h AvI,IvA,EvI,IvE 2>&1 | grep -v amdgpu > gpurun_out/s22/times.txt; cat gpurun_out/s22/times.txt
o=gpurun_out/s22/ablate.txt; : > $o
root=$(pwd)
for d in 0 16 3 31; do
  export TUNE=assemble_stream_dbg=$d
  echo "== dbg=$d" >> $o
  bash scratch/prof_asm.sh abl$d $root/scratch/time_assembly.py a1h AvI | grep -E "k_sa_emit|k_sa_rows1|k_sa_flags|k_sa_pairs" >> $o
done
cat $o
