#!/bin/bash
# config 5's applies on this round's kernels + the empty-launch floor under rocprofv3
root=$(pwd); out=$root/gpurun_out/r04b; mkdir -p $out
export TMPDIR=/tmp
(cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $out/empty_kt -- $root/scratch/empty_bench > $out/empty_kt.log 2>&1)
python3 scratch/rocsum.py $out/empty_kt "k_empty" > $out/empty_launch_rocprof.txt; cat $out/empty_launch_rocprof.txt | head -20; grep EMPTY $out/empty_kt.log
for spec in "AvI 16" "IvA 16" "EvI 16" "AvI 128" "EvI 128" "IvA 128"; do
  set -- $spec
  timeout -k 10 600 python3 bench.py --config a1h --matrix $1 --fields $2 --steps 32 --warmup 8 --repeats 5 --no-cpu-baseline > $out/bench_a1h_$1_$2f.json.log 2>$out/bench_a1h_$1_$2f.err
  echo "$spec rc=$?"; python3 - <<PY
import json
try:
    d=json.loads([l for l in open("$out/bench_a1h_$1_$2f.json.log") if l.startswith("{")][-1])
    r=d["roofline"]; print("$spec", d["config"]["kernel"], "kernel_us %.1f frac %.3f single %s value %.3e" % (r["kernel_us"], r["frac"], r.get("single_launch",{}).get("frac"), d["value"]))
except Exception as e: print("$spec failed", e)
PY
done
find $out -name "*.csv" -size +4M -delete
