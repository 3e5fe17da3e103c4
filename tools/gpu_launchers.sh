#!/bin/bash
# the two ways the driver starts the bench, on the one GPU of the box
set -u
export TMPDIR=/tmp
echo "== torch.distributed.run, 1 rank"; timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline 2>&1 | grep '^{' | cut -c1-400; echo "rc=${PIPESTATUS[0]}"
echo "== torch.distributed.run, 1 rank, forced N>1 code path"; timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29512 bench.py --gpus 1 --force-dist --steps 20 --warmup 5 --no-cpu-baseline 2>&1 | grep '^{' | cut -c1-600; echo "rc=${PIPESTATUS[0]}"
echo "== bench.py --gpus 2 on a one-GPU box (must fail loudly, non-zero)"; timeout 600 python bench.py --gpus 2 --steps 20 --warmup 5 --no-cpu-baseline 2>&1 | tail -5 | cut -c1-300; echo "rc=${PIPESTATUS[0]}"
