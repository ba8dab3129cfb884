#!/bin/bash
mkdir -p gpurun_out/r05s15; o=gpurun_out/r05s15; : > $o/t.txt
for t in "" assemble_stream=1; do
 echo "== TUNE=[$t]" >> $o/t.txt
 TUNE=$t timeout -k 10 300 python scratch/time_assembly.py g5,g20 AvI,IvA,EvI,IvE,XvE >> $o/t.txt 2>&1
done
grep -v amdgpu.ids $o/t.txt
