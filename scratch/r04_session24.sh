#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/s24
o=gpurun_out/s24/emit.txt; : > $o
for t in "" "assemble_stream_emit_cpt=1" "assemble_stream_emit_cpt=1,assemble_stream_emit_blocks=4096" "assemble_stream_emit_cpt=1,assemble_stream_emit_blocks=8192" "assemble_stream_emit_cpt=1,assemble_stream_emit_blocks=16384" "assemble_stream_emit_cpt=2,assemble_stream_emit_blocks=8192" "assemble_stream_emit_cpt=2,assemble_stream_emit_blocks=16384"; do
  echo "== $t" >> $o
  TUNE=$t python scratch/time_assembly.py a1h AvI,IvA,EvI,IvE 2>&1 | grep -v amdgpu >> $o
  TUNE=$t python scratch/time_assembly.py g1 AvI,IvA,EvI,IvE 2>&1 | grep -v amdgpu >> $o
done
cat $o
