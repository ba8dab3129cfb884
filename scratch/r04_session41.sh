#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/s43
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "g1 or config4 or config5 or shortrow or IvA or IvE or capture or graph or prepare or smooth or apply" > gpurun_out/s43/pytest.log 2>&1
rc=$?; echo "pytest rc=$rc"; tail -4 gpurun_out/s43/pytest.log
[ $rc -eq 0 ] || exit 1
o=gpurun_out/s43/ivx.txt; : > $o
for m in IvA IvE; do python scratch/kernel_choice.py a1h $m 16,128 auto 2>&1 | grep nf= >> $o; done
python scratch/kernel_choice.py g1 IvA 16,64 auto 2>&1 | grep nf= >> $o
python scratch/kernel_choice.py g1 IvE 16,64 auto 2>&1 | grep nf= >> $o
python scratch/kernel_choice.py g5 IvE 16,64 auto 2>&1 | grep nf= >> $o
cat $o
python scratch/irow_probe2.py 2>&1 | grep -v amdgpu | tee gpurun_out/s43/probe2.txt
