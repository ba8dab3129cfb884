#!/bin/bash
# Round-2 measurement set (GPU box, from the repo root).  Outputs under gpurun_out/r02/, copied to profiles/ by collect_profiles.py
out=gpurun_out/r02; mkdir -p $out
export TMPDIR=/tmp
timeout -k 10 500 scratch/prof_r02.sh > $out/prof_r02.log 2>&1; echo "headline profile: rc $?"
timeout -k 10 400 python scratch/time_assembly.py g20,g5,g1,a1h AvI,IvA,EvI,IvE,EvA 2>&1 | grep -v amdgpu > $out/assembly_times.txt; echo "assembly times: rc $?"
timeout -k 10 300 python scratch/coupler_step.py g20,g5,g1 2>&1 | grep -v amdgpu > $out/coupler_step.txt; echo "coupler: rc $?"
DIV=6 timeout -k 10 300 scratch/prof_asm.sh a1hAvI $PWD/scratch/time_assembly.py a1h AvI > $out/assembly_a1h_AvI_kernels.txt 2>&1
DIV=6 timeout -k 10 300 scratch/prof_asm.sh a1hEvI $PWD/scratch/time_assembly.py a1h EvI > $out/assembly_a1h_EvI_kernels.txt 2>&1
DIV=6 timeout -k 10 300 scratch/prof_asm.sh a1hIvE $PWD/scratch/time_assembly.py a1h IvE > $out/assembly_a1h_IvE_kernels.txt 2>&1
timeout -k 10 500 python scratch/apply_all.py g5,g1 2>&1 | grep -v amdgpu > $out/apply_all_matrices.txt; echo "apply all: rc $?"
timeout -k 10 300 python scratch/evi_apply.py g1 2>&1 | grep -v amdgpu > $out/evi_apply_g1.txt
timeout -k 10 200 python scratch/evi_apply.py g5 2>&1 | grep -v amdgpu > $out/evi_apply_g5.txt
timeout -k 10 400 scratch/prof_evi.sh > $out/prof_evi.log 2>&1; cp gpurun_out/prof_evi/summary.txt $out/evi_g1_colsweep_profile.txt; cp gpurun_out/prof_evi/summary.json $out/evi_g1_colsweep_profile.json
timeout -k 10 200 python scratch/time_smooth.py 2>&1 | grep -v amdgpu > $out/smoothing_times.txt
timeout -k 10 200 python bench.py --config g1 --no-cpu-baseline --steps 96 --warmup 32 > $out/bench_g1_AvI_64f.json.log 2>/dev/null
timeout -k 10 200 python bench.py --config g1 --matrix IvA --no-cpu-baseline --steps 96 --warmup 32 > $out/bench_g1_IvA_64f.json.log 2>/dev/null
timeout -k 10 200 python bench.py --queue-depth 1 --no-cpu-baseline > $out/bench_g5_AvI_64f_depth1.json.log 2>/dev/null
timeout -k 10 200 python bench.py --variants --no-cpu-baseline > $out/bench_g5_AvI_64f_with_variants.json.log 2>/dev/null
timeout -k 10 300 scratch/run_bench_dist1.sh 2>&1 | grep "^{" > $out/bench_torchrun_1rank.json.log
timeout -k 10 300 scratch/run_bench_dist2_gloo.sh 2>&1 | grep "^{" > $out/bench_torchrun_2ranks_gloo_rehearsal.json.log
ls -la $out
