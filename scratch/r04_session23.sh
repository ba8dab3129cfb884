#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/s23
python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "streamed_build or random_sorted_grids" > gpurun_out/s23/pytest.log 2>&1
echo "pytest rc=$?"; tail -3 gpurun_out/s23/pytest.log
o=gpurun_out/s23/emit.txt; : > $o
for t in "" assemble_stream_emit_blocks=2048 assemble_stream_emit_blocks=4096 assemble_stream_emit_blocks=8192 assemble_stream_emit_cpt=4 "assemble_stream_emit_cpt=4,assemble_stream_emit_blocks=2048" "assemble_stream_emit_cpt=1,assemble_stream_emit_blocks=4096"; do
  echo "== $t" >> $o
  TUNE=$t python scratch/time_assembly.py a1h AvI,IvA,EvI,IvE 2>&1 | grep -v amdgpu >> $o
done
cat $o
