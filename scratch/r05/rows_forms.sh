#!/bin/bash
for t in "" "assemble_stream_rowsl=0,assemble_stream_rows4=1" "assemble_stream_rowsl=0,assemble_stream_rows4=0" "assemble_stream_rowsl=1"; do
echo "== TUNE=$t"
TUNE=$t python3 scratch/time_assembly.py a1h AvI,EvI,IvA,IvE,XvE 2>&1 | grep -v "amdgpu\|regrid_matrices"
done
for t in "" "assemble_stream_rows4=1" "assemble_stream_rows4=0"; do
echo "== TUNE=$t"
TUNE=$t python3 scratch/time_assembly.py g1 AvI,EvI,IvA,IvE 2>&1 | grep -v "amdgpu\|regrid_matrices"
done
