#!/bin/bash
# PMC traffic of the Antarctic builds: separate FETCH_SIZE / WRITE_SIZE passes (no kernel trace in the same run)
root=$(pwd); out=$root/gpurun_out/r04pmc; mkdir -p $out
export TMPDIR=/tmp
for m in ${@:-AvI IvE}; do
  for c in FETCH_SIZE WRITE_SIZE; do
    (cd /tmp && rocprofv3 --pmc $c --output-format csv -d $out/pmc_${m}_$c -- python3 $root/scratch/time_assembly.py a1h $m > $out/pmc_${m}_$c.log 2>&1) || exit 1
  done
  for c in FETCH_SIZE WRITE_SIZE; do python3 scratch/rocsum.py $out/pmc_${m}_$c k_; done > $out/assembly_a1h_${m}_pmc.txt
  cat $out/assembly_a1h_${m}_pmc.txt
done
find $out -name "*.csv" -size +4M -delete
