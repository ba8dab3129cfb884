#!/bin/bash
# bench.py's N>1 code path with ONE rank under torchrun, gathers through the library's own RCCL calls (ICEBIN_BENCH_SHARDED=cabi)
set -e
export ICEBIN_BENCH_SHARDED=cabi
python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 1 --steps 64 --warmup 32 --repeats 5 --no-cpu-baseline
python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29534 bench.py --gpus 1 --steps 16 --warmup 4 --repeats 3 --no-cpu-baseline --config g1 --matrix IvA --fields-total 64
