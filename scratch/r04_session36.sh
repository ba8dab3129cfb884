#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/s36
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "g1 or config4 or config5 or shortrow or IvA or IvE or capture or graph or prepare" > gpurun_out/s36/pytest.log 2>&1
rc=$?; echo "pytest rc=$rc"; tail -4 gpurun_out/s36/pytest.log
[ $rc -eq 0 ] || exit 1
o=gpurun_out/s36/ivx.txt; : > $o
for t in shortrow_deadbits=0 shortrow_deadbits=1; do
  for m in IvA IvE; do
    TUNE=$t python scratch/kernel_choice.py a1h $m 16,128 auto 2>&1 | grep nf= | sed "s/^/$t /" >> $o
  done
  TUNE=$t python scratch/kernel_choice.py g1 IvA 64 auto 2>&1 | grep nf= | sed "s/^/$t /" >> $o
done
cat $o
