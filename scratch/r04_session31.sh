#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/s31
bash scratch/prof_asm.sh emc $(pwd)/scratch/time_assembly.py a1h AvI,IvE > gpurun_out/s31/kernels.txt
grep -E "k_sa_flags|k_em_classes|k_sa_emit" gpurun_out/s31/kernels.txt
