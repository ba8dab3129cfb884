#!/bin/bash
# rehearsal of the N>1 bench path with one rank under torchrun (weak and strong modes)
set -e
python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 1 --steps 320 --warmup 32 --no-cpu-baseline
python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29534 bench.py --gpus 1 --steps 64 --warmup 16 --no-cpu-baseline --config g1 --fields-total 64
python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29535 bench.py --gpus 1 --steps 64 --warmup 16 --no-cpu-baseline --config g1 --fields-total 64 --matrix IvA
