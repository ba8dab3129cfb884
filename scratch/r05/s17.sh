#!/bin/bash
mkdir -p gpurun_out/r05s17; o=gpurun_out/r05s17
timeout -k 10 300 python scratch/chain3.py 16 g5 > $o/chain.txt 2>&1; echo "rc $?" >> $o/chain.txt
cat $o/chain.txt
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "fused_pair" > $o/pytest.txt 2>&1; echo "pytest rc $?" >> $o/pytest.txt
tail -15 $o/pytest.txt
