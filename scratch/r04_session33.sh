#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/s33
timeout -k 10 1000 python scratch/fuzz_builds.py 1000 3000 2>&1 | grep -v amdgpu > gpurun_out/s33/fuzz_builds.log
echo rc=$?; grep -c ok gpurun_out/s33/fuzz_builds.log; grep -E "MISMATCH|mismatches|Error|Traceback" gpurun_out/s33/fuzz_builds.log | head -20; tail -3 gpurun_out/s33/fuzz_builds.log
