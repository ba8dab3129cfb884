#!/bin/bash
out=gpurun_out/r05s33; mkdir -p $out
timeout -k 10 1000 python scratch/fuzz_sharded.py 0 300 15 > $out/sharded.log 2>&1; echo "rc $?" >> $out/sharded.log
grep "^seeds\|^batches" $out/sharded.log | cut -c1-400
