#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/s28
timeout -k 10 400 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "streamed_build or random_sorted_grids" > gpurun_out/s28/pytest.log 2>&1
rc=$?; echo "pytest rc=$rc"; tail -15 gpurun_out/s28/pytest.log
[ $rc -eq 0 ] || exit 1
o=gpurun_out/s28/rows.txt; : > $o
for t in "" "assemble_stream_rowsl=1" "assemble_stream_rowsl=1,assemble_stream_rowsl_r=8" "assemble_stream_rowsl=1,assemble_stream_rowsl_r=4"; do
  echo "== $t" >> $o
  TUNE=$t python scratch/time_assembly.py a1h AvI,IvA,EvI,IvE 2>&1 | grep -v amdgpu >> $o
done
export TUNE=assemble_stream_rowsl=1
bash scratch/prof_asm.sh rl3 $(pwd)/scratch/time_assembly.py a1h AvI,EvI,IvE | grep -E "k_sa_rows" >> $o
cat $o
