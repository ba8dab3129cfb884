#!/bin/bash
mkdir -p gpurun_out/r05s16; o=gpurun_out/r05s16
timeout -k 10 1100 python -m pytest tests/test_distributed_gloo.py -x -q -m gpu -k "sharded_assembly" > $o/pytest.txt 2>&1; echo "pytest rc $?" >> $o/pytest.txt
tail -25 $o/pytest.txt
