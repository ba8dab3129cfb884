#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/s18
python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "streamed_build or random_sorted_grids or full_size_assembly" > gpurun_out/s18/pytest_stream.log 2>&1
echo "pytest stream rc=$?"; tail -5 gpurun_out/s18/pytest_stream.log
timeout -k 10 600 python -m pytest tests/test_distributed_gloo.py -x -q -m gpu -k sharded_assembly > gpurun_out/s18/pytest_dist.log 2>&1
echo "pytest dist rc=$?"; tail -3 gpurun_out/s18/pytest_dist.log
python scratch/time_assembly.py a1h AvI,IvA,EvI,IvE,AvX,XvA > gpurun_out/s18/asm_stream.txt 2>&1; cat gpurun_out/s18/asm_stream.txt
python scratch/time_assembly.py g1,g1h AvI,IvA > gpurun_out/s18/asm_g1.txt 2>&1; cat gpurun_out/s18/asm_g1.txt
for m in AvI IvA; do
  DIV=6 bash scratch/prof_asm.sh r04h_$m $(pwd)/scratch/time_assembly.py a1h $m > gpurun_out/s18/kern_$m.txt 2>&1 || exit 1
  grep "k_sa\|k_fa\|k_ms" gpurun_out/s18/kern_$m.txt | head -10
done
bash scratch/r04_pmc_asm.sh AvI > gpurun_out/s18/pmc.log 2>&1
python3 - <<PY
import re
tot={"FETCH_SIZE":0,"WRITE_SIZE":0}
for l in open("gpurun_out/r04pmc/assembly_a1h_AvI_pmc.txt"):
    mo=re.search(r"^(\S.*?)\s+(FETCH_SIZE|WRITE_SIZE)\s+dispatches\s+(\d+)\s+mean\s+([\d.]+)",l)
    if mo and int(mo.group(3))%6==0: tot[mo.group(2)]+=float(mo.group(4))*(int(mo.group(3))//6)
print("AvI",tot,"total %.0f MB = %.2f x B_asm"%(sum(tot.values()),sum(tot.values())/1250.9))
PY
