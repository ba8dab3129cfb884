#!/bin/bash
mkdir -p gpurun_out/r05s25; o=gpurun_out/r05s25; : > $o/t.txt
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "shortrow or shared_col or irow or IvE or apply" > $o/pytest.txt 2>&1; echo "pytest rc $?" >> $o/pytest.txt
tail -3 $o/pytest.txt
for m in IvA IvE; do timeout -k 10 300 python scratch/kernel_choice.py a1h $m 16,128 auto 2>&1 | grep nf= >> $o/t.txt; done
timeout -k 10 300 python scratch/depth1.py "g1:IvE:64:" "g1:IvA:64:" "g5:IvE:64:" "g5:IvE:16:" 2>&1 | grep -v amdgpu >> $o/t.txt
cat $o/t.txt
