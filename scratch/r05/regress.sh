#!/bin/bash
out=gpurun_out/r05regress; mkdir -p $out
timeout -k 10 900 python -m pytest tests -q -x -m gpu -k "not config5 and not config4" > $out/pytest.txt 2>&1; echo "pytest rc $?" >> $out/pytest.txt
tail -n 3 $out/pytest.txt
timeout -k 10 400 python scratch/fuzz_builds.py 900 120 > $out/builds.log 2>&1; echo "rc $?" >> $out/builds.log
tail -n 2 $out/builds.log
timeout -k 10 600 python scratch/fuzz_sharded.py 400 60 15 > $out/sharded.log 2>&1; echo "rc $?" >> $out/sharded.log
grep "^batches\|^rc" $out/sharded.log
