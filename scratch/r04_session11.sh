#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/s11
python scratch/time_assembly.py a1h AvI,IvA,EvI,IvE,AvX,XvE > gpurun_out/s11/asm_stream.txt 2>&1; cat gpurun_out/s11/asm_stream.txt
timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29513 bench.py --gpus 1 --steps 64 --warmup 32 --repeats 5 --no-cpu-baseline 2>gpurun_out/s11/dist1.err | grep "^{" > gpurun_out/s11/bench_torchrun_1rank.json.log
python3 - <<PY
import json
d=json.loads(open("gpurun_out/s11/bench_torchrun_1rank.json.log").read().splitlines()[-1])
print("gather_via:", d.get("gather_via"), "| fallback:", d.get("gather_via_fallback"), "| check:", d.get("gather_check",{}).get("pass"), "| asm:", d["assembly"].get("sharded"))
PY
bash scratch/r04_pmc_asm.sh AvI IvE > gpurun_out/s11/pmc.log 2>&1
grep "k_sa\|k_fa_pelem" gpurun_out/r04pmc/assembly_a1h_AvI_pmc.txt gpurun_out/r04pmc/assembly_a1h_IvE_pmc.txt
