#!/bin/bash
mkdir -p gpurun_out/r05s14; o=gpurun_out/r05s14
timeout -k 10 900 python -m pytest tests/test_distributed_gloo.py -x -q -m gpu > $o/pytest.txt 2>&1; echo "pytest rc $?" >> $o/pytest.txt
tail -15 $o/pytest.txt
