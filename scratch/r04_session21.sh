#!/bin/bash
# ablation of the streamed build's emit pass (a1h AvI): which part of it costs the time
set -o pipefail
mkdir -p gpurun_out/s21
o=gpurun_out/s21/ablate.txt; : > $o
root=$(pwd)
for d in 0 1 2 3 4 8 16 31; do
  export TUNE=assemble_stream_dbg=$d
  echo "== dbg=$d" >> $o
  bash scratch/prof_asm.sh abl$d $root/scratch/time_assembly.py a1h AvI | grep -E "k_sa_emit|k_sa_rows1|k_sa_flags" >> $o
done
cat $o
