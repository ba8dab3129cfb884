#!/bin/bash
mkdir -p gpurun_out/s34
python bench.py > gpurun_out/s34/bench_default.json 2> gpurun_out/s34/bench_default.err; echo "bench rc=$?"
python bench.py --config a1h --matrix IvA --fields 16 --steps 32 --warmup 8 --repeats 3 --no-cpu-baseline > gpurun_out/s34/bench_a1h_IvA.json 2>/dev/null
python - <<'PY'
import json
for f in ("bench_default", "bench_a1h_IvA"):
    d = json.loads(open("gpurun_out/s34/%s.json" % f).read().strip().splitlines()[-1])
    r = d["roofline"]
    print(f, "frac", round(r["frac"], 3), "achieved", round(r["achieved"]), r.get("measured_streams"))
PY
