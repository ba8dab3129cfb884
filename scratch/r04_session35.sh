#!/bin/bash
mkdir -p gpurun_out/r04
./scratch/store_bench 17598064 16 > gpurun_out/r04/store_ceiling.txt 2>&1
./scratch/store_bench 17598064 128 >> gpurun_out/r04/store_ceiling.txt 2>&1
timeout -k 10 300 python3 bench.py > gpurun_out/r04/bench_g5_AvI_64f_default.json.log 2>/dev/null; echo "bench default rc $?"
timeout -k 10 300 python3 bench.py --steps 20 --warmup 5 > gpurun_out/r04/bench_g5_AvI_64f_driver20.json.log 2>/dev/null; echo "bench driver20 rc $?"
for spec in "AvI 16" "IvA 16" "EvI 16"; do set -- $spec
  timeout -k 10 600 python3 bench.py --config a1h --matrix $1 --fields $2 --steps 32 --warmup 8 --repeats 5 --no-cpu-baseline > gpurun_out/r04/bench_a1h_$1_$2f.json.log 2>/dev/null; echo "a1h $spec rc=$?"
done
python3 - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r04/bench_*64f_d*.json.log")+glob.glob("gpurun_out/r04/bench_a1h_*_16f.json.log")):
    d=json.loads(open(f).read().strip().splitlines()[-1]); r=d["roofline"]
    print(f.split("/")[-1], round(r["frac"],3), {k:(round(v) if isinstance(v,float) and v>10 else v) for k,v in (r.get("measured_streams") or {}).items() if k!="what"})
PY
