#!/bin/bash
# rehearsal of bench.py's N=2 path on ONE GPU: two ranks share the card over gloo (RCCL refuses two ranks on one device);
# exercises field sharding, the grouped all-gather choreography and the max-over-ranks timing -- not a measurement
set -e
export ICEBIN_BENCH_BACKEND=gloo
python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29543 bench.py --gpus 2 --steps 64 --warmup 8 --no-cpu-baseline
python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29544 bench.py --gpus 2 --steps 16 --warmup 4 --no-cpu-baseline --config g1 --fields-total 64
python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29545 bench.py --gpus 2 --steps 8 --warmup 2 --no-cpu-baseline --config g1 --fields-total 64 --matrix IvA
