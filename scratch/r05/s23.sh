#!/bin/bash
mkdir -p gpurun_out/r05s23; o=gpurun_out/r05s23; : > $o/t.txt
timeout -k 10 1000 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "assembly or fast or stream or random or config5 or identity or dims" > $o/pytest.txt 2>&1; echo "pytest rc $?" >> $o/pytest.txt
tail -4 $o/pytest.txt
for t in "scan_chained=0" ""; do
 echo "== TUNE=[$t]" >> $o/t.txt
 TUNE=$t timeout -k 10 300 python scratch/time_assembly.py g20,g5,g1 AvI,IvA,EvI,IvE,XvE 2>&1 | grep -v amdgpu >> $o/t.txt
done
cat $o/t.txt
