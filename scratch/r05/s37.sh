#!/bin/bash
out=gpurun_out/r05s37; mkdir -p $out
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -q -x -k "not config5 and not config4" > $out/pytest.txt 2>&1; echo "pytest rc $?" >> $out/pytest.txt
tail -n 3 $out/pytest.txt
timeout -k 10 300 python scratch/fuzz_builds.py 700 60 > $out/builds.log 2>&1; echo "rc $?" >> $out/builds.log
tail -n 2 $out/builds.log
for m in AvI EvI IvE IvA XvE; do
for sh in -1 0 1 2; do
TUNE=assemble_range_shape=$sh python3 scratch/r05/build_trace.py g5 $m 2>&1 | grep "per build" | sed "s/$/ shape $sh/"
done; done
