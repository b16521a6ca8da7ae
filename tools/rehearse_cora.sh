#!/bin/bash
set -o pipefail
out=gpurun_out/rehearse_cora
mkdir -p "$out"
port=29740
for k in 1 2 3; do
  C="--workload cora --steps $k --warmup 0 --no-alt --cpu-budget 0"
  timeout -k 10 200 python bench.py $C > "$out/n1_k$k.log" 2>&1 || exit 1
  for n in 2 3; do
    port=$((port + 1))
    DGNN_PIPELINE=bcast DGNN_FAULT_TIMEOUT=150 timeout -k 10 240 python -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 \
        --master-port $port bench.py --gpus $n --backend gloo $C > "$out/n${n}_k$k.log" 2>&1 || { echo "N=$n k=$k FAILED"; exit 2; }
  done
done
