#!/bin/bash
./scratch/r05/addchain
python3 scratch/time_assembly.py a1h,g1 AvI,EvI,IvA,IvE 2>&1 | grep -v amdgpu
python -m pytest tests/test_gpu_parity.py -q -x -k "config5_full_size or streamed" 2>&1 | tail -3
