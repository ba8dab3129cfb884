#!/bin/bash
# Round-5 measurement set (GPU box, from the repo root).  Outputs under gpurun_out/r05/; scratch/copy_r05.py files them under profiles/.
# usage: collect_r05.sh [part ...]   parts: headline pmc asm a1h table misc dist tests   (default: all)
root=$(pwd); out=$root/gpurun_out/r05; mkdir -p $out
export TMPDIR=/tmp
parts=${@:-headline pmc asm a1h table misc dist tests}
has() { [[ " $parts " == *" $1 "* ]]; }
rp() { tag=$1; shift; (cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $out/$tag -- "$@" > $out/$tag.log 2>&1); }
pmc() { tag=$1; shift; for c in FETCH_SIZE WRITE_SIZE; do (cd /tmp && rocprofv3 --pmc $c --output-format csv -d $out/${tag}_$c -- "$@" > $out/${tag}_$c.log 2>&1); done; }
if has tests; then
  python3 -m pytest tests -q -m gpu 2>&1 | tail -4 > $out/pytest_gpu.txt; cat $out/pytest_gpu.txt
fi
if has headline; then
  timeout -k 10 300 python3 bench.py > $out/bench_g5_AvI_64f_default.json.log 2>/dev/null; echo "bench default rc $?"
  timeout -k 10 300 python3 bench.py --steps 20 --warmup 5 > $out/bench_g5_AvI_64f_driver20.json.log 2>/dev/null; echo "bench driver20 rc $?"
  rp kt_default python3 $root/bench.py --no-cpu-baseline --no-extras
  rp kt_driver20 python3 $root/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras
  rp kt_depth1 python3 $root/bench.py --queue-depth 1 --steps 320 --warmup 32 --repeats 5 --no-cpu-baseline --no-extras
  rp kt_depth1_plain python3 $root/bench.py --queue-depth 1 --steps 320 --warmup 32 --repeats 5 --no-cpu-baseline --no-extras --no-kernel-events
  for t in kt_default kt_driver20 kt_depth1 kt_depth1_plain; do python3 scratch/rocsum.py $out/$t spmm_ --json $out/$t.summary.json > $out/$t.summary.txt; grep "^{" $out/$t.log > $out/$t.bench.json; done
  # the floor of one launch: an empty kernel under the kernel trace (plain launches and launches with HIP events attached) and by HIP events alone
  rp empty_kt $root/scratch/empty_bench
  python3 scratch/rocsum.py $out/empty_kt k_empty > $out/empty_launch_rocprof.txt
  python3 - >> $out/empty_launch_rocprof.txt <<PY
import csv,glob,statistics as st
f=glob.glob("$out/empty_kt/**/*kernel_trace.csv",recursive=True)[0]
by={}
for r in csv.DictReader(open(f)): by.setdefault(r["Kernel_Name"],[]).append((int(r["Start_Timestamp"]),int(r["End_Timestamp"])))
print("# the same trace split: the first 20 launches of every kernel are PLAIN (hipLaunchKernelGGL), the other 200 carry HIP events (hipExtLaunchKernelGGL)")
for k,v in by.items():
    v.sort(); d=[(e-s)/1e3 for s,e in v]
    print("%-22s plain: median %.2f min %.2f max %.2f us | with events: median %.2f min %.2f max %.2f us" % (k[:22], st.median(d[:20]),min(d[:20]),max(d[:20]),st.median(d[20:]),min(d[20:]),max(d[20:])))
PY
  grep EMPTY $out/empty_kt.log >> $out/empty_launch_rocprof.txt; echo "# without the profiler:" >> $out/empty_launch_rocprof.txt; ./scratch/empty_bench >> $out/empty_launch_rocprof.txt 2>&1
fi
if has pmc; then
  pmc pmc_default python3 $root/bench.py --steps 64 --warmup 32 --repeats 3 --no-cpu-baseline --no-extras
  pmc pmc_depth1 python3 $root/bench.py --queue-depth 1 --steps 64 --warmup 32 --repeats 3 --no-cpu-baseline --no-extras
  for t in pmc_default pmc_depth1; do for c in FETCH_SIZE WRITE_SIZE; do python3 scratch/rocsum.py $out/${t}_$c spmm_ --json $out/${t}_$c.json; done > $out/$t.summary.txt; done
fi
if has asm; then
  timeout -k 10 600 python3 scratch/time_assembly.py g20,g5,g1,g1h,a1h AvI,IvA,EvI,IvE,AvX,XvE,EvA 2>&1 | grep -v amdgpu > $out/assembly_times.txt; echo "assembly times rc $?"
  TUNE=assemble_stream=0 timeout -k 10 600 python3 scratch/time_assembly.py g1,a1h AvI,IvA,EvI,IvE 2>&1 | grep -v amdgpu > $out/assembly_times_per_range_kernels.txt
  timeout -k 10 300 python3 scratch/coupler_step.py g20,g5,g1 2>&1 | grep -v amdgpu > $out/coupler_step.txt
  for m in AvI IvE EvI IvA; do
    rp asm_a1h_$m python3 $root/scratch/time_assembly.py a1h $m; python3 scratch/rocsum.py $out/asm_a1h_$m k_ > $out/assembly_a1h_${m}_kernels.txt
  done
  for m in AvI IvE; do
    pmc pmc_asm_a1h_$m python3 $root/scratch/time_assembly.py a1h $m
    for c in FETCH_SIZE WRITE_SIZE; do python3 scratch/rocsum.py $out/pmc_asm_a1h_${m}_$c k_; done > $out/assembly_a1h_${m}_pmc.txt
  done
fi
if has a1h; then
  for spec in "AvI 16" "IvA 16" "EvI 16" "AvI 128" "EvI 128" "IvA 128"; do
    set -- $spec
    timeout -k 10 600 python3 bench.py --config a1h --matrix $1 --fields $2 --steps 32 --warmup 8 --repeats 5 --no-cpu-baseline > $out/bench_a1h_$1_$2f.json.log 2>/dev/null; echo "a1h $spec rc=$?"
  done
  timeout -k 10 600 python3 scratch/kernel_choice.py a1h AvI 16,128 auto 2>&1 | grep nf= > $out/apply_a1h_one_launch.txt
  timeout -k 10 600 python3 scratch/kernel_choice.py a1h IvA 16,128 auto 2>&1 | grep nf= >> $out/apply_a1h_one_launch.txt
  timeout -k 10 600 python3 scratch/kernel_choice.py a1h IvE 16,128 auto 2>&1 | grep nf= >> $out/apply_a1h_one_launch.txt
  timeout -k 10 600 python3 scratch/kernel_choice.py a1h EvI 16,128 auto,rowblock,colsweep,rowgroup 2>&1 | grep nf= >> $out/apply_a1h_one_launch.txt
fi
if has evi; then
  # the tiled row groups (spmm_grouptile_kernel) against the LDS-atomic form: one apply per launch, kernel-event durations; rocprofv3
  # kernel stats + SQ counters + FETCH / WRITE passes of both forms at 1 km, 64 fields
  timeout -k 10 500 python3 scratch/depth1.py "g5:EvI:16:rowgroup_form=0" "g5:EvI:16:rowgroup_form=1" "g5:EvI:64:rowgroup_form=0" "g5:EvI:64:rowgroup_form=1" \
     "g1:EvI:64:rowgroup_form=0" "g1:EvI:64:rowgroup_form=1" "g1:EvI:64:rowgroup_form=1,grouptile_seg=128" "g1:EvI:16:rowgroup_form=0,kernel=rowgroup" "g1:EvI:16:rowgroup_form=1,kernel=rowgroup" 2>&1 | grep -v amdgpu > $out/evi_rowgroup_forms.txt
  for t in rowgroup_form=0 rowgroup_form=1 "rowgroup_form=1,grouptile_seg=256"; do
    TUNE=$t timeout -k 10 400 python3 scratch/kernel_choice.py a1h EvI 16,128 rowgroup 2>&1 | grep nf= | sed "s/^/[$t] /" >> $out/evi_rowgroup_forms.txt
  done
  bash scratch/r05/pmc.sh g1_atomics g1 64 rowgroup_form=0 > /dev/null 2>&1; bash scratch/r05/pmc.sh g1_tiles g1 64 rowgroup_form=1 > /dev/null 2>&1
  for t in g1_atomics g1_tiles; do grep -v "elementwise\|rocprofv3" gpurun_out/r05pmc/$t/summary.txt | sed "s/void ibh:://" > $out/pmc_evi_$t.summary.txt; grep "elementwise.*FETCH_SIZE\|elementwise.*WRITE_SIZE" gpurun_out/r05pmc/$t/summary.txt | head -2 >> $out/pmc_evi_$t.summary.txt; done
fi
if has table; then
  timeout -k 10 600 python3 scratch/apply_table.py g5,g1 2>&1 | grep -v amdgpu > $out/apply_all_matrices.txt; echo "table rc $?"
  timeout -k 10 200 python3 scratch/chain3.py 16 2>&1 | grep -v amdgpu > $out/config3_chain.txt; timeout -k 10 200 python3 scratch/chain3.py 64 2>&1 | grep -v amdgpu >> $out/config3_chain.txt
fi
if has misc; then
  timeout -k 10 200 python3 bench.py --config g1 --no-cpu-baseline --steps 96 --warmup 32 --repeats 5 > $out/bench_g1_AvI_64f.json.log 2>/dev/null
  timeout -k 10 200 python3 bench.py --config g1 --matrix IvA --no-cpu-baseline --steps 96 --warmup 32 --repeats 5 > $out/bench_g1_IvA_64f.json.log 2>/dev/null
  timeout -k 10 200 python3 bench.py --fields 1 --no-cpu-baseline --steps 320 --warmup 32 > $out/bench_g5_AvI_1f_config2.json.log 2>/dev/null
  timeout -k 10 200 python3 bench.py --fields 1 --matrix IvA --no-cpu-baseline --steps 320 --warmup 32 > $out/bench_g5_IvA_1f_config2.json.log 2>/dev/null
fi
if has dist; then
  timeout -k 10 300 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 1 --steps 64 --warmup 32 --repeats 5 --no-cpu-baseline 2>/dev/null | grep "^{" > $out/bench_torchrun_1rank_cabi_default.json.log
  ICEBIN_BENCH_SHARDED=torch timeout -k 10 300 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29534 bench.py --gpus 1 --steps 64 --warmup 32 --repeats 5 --no-cpu-baseline 2>/dev/null | grep "^{" > $out/bench_torchrun_1rank_torch.json.log
  timeout -k 10 300 scratch/run_bench_dist2_gloo.sh 2>&1 | grep "^{" > $out/bench_torchrun_2ranks_gloo_rehearsal.json.log
fi
find $out -name "*.csv" -size +4M -delete      # (dispatch-level traces of the long runs stay on the box; the summaries travel)
ls $out | head -100
