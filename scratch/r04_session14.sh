#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/s14
python -m pytest tests -x -q -m gpu > gpurun_out/s14/pytest_gpu.log 2>&1
echo "pytest rc=$?"; tail -6 gpurun_out/s14/pytest_gpu.log
python scratch/kernel_choice.py a1h EvI 16,128 auto > gpurun_out/s14/kc.txt 2>&1; python scratch/kernel_choice.py a1h AvI 16,128 auto >> gpurun_out/s14/kc.txt 2>&1; python scratch/kernel_choice.py a1h IvA 16,128 auto >> gpurun_out/s14/kc.txt 2>&1; grep nf= gpurun_out/s14/kc.txt
