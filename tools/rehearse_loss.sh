#!/bin/bash
# Loss over all training nodes after k optimiser steps, k = 0, 1, 2, at N = 1, 2, 4 ranks sharing one GPU
# (host-staged gloo): the first checks the forward pass, the next two the backward pass and the update.
set -o pipefail
scale=${1:-0.3}
out=gpurun_out/rehearse_loss
mkdir -p "$out"
port=29640
for k in 1 2 3; do
  common="--scale $scale --steps $k --warmup 0 --no-alt --cpu-budget 0"
  timeout -k 10 200 python bench.py $common > "$out/n1_k$k.log" 2>&1 || exit 1
  for n in 2 4; do
    port=$((port + 1))
    DGNN_FAULT_TIMEOUT=150 timeout -k 10 240 python -m torch.distributed.run --nnodes=1 --nproc-per-node $n \
        --master-addr 127.0.0.1 --master-port $port bench.py --gpus $n --backend gloo $common > "$out/n${n}_k$k.log" 2>&1 || { echo "N=$n k=$k FAILED"; exit 2; }
  done
  echo "k=$k done"
done
