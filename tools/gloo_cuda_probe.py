#!/usr/bin/env python3
"""Is a stall of the gloo-with-GPU rehearsal (bench.py --backend gloo) ours or gloo's? The same exchange pattern as
StagePipeline's bcast form -- async rooted broadcasts of CUDA tensors with two rotating buffers, a matrix product
per stage -- with plain torch ops only (no dgnn code). Run under torch.distributed.run with 2 ranks on one GPU."""
import os
import sys
import time

import torch
import torch.distributed as dist

rank, world = int(os.environ['RANK']), int(os.environ['WORLD_SIZE'])
torch.cuda.set_device(0)
dist.init_process_group('gloo', rank=rank, world_size=world)
dev = torch.device('cuda', 0)
s, F = 23297, 608
own = torch.rand(s, F, device=dev)
W = torch.rand(F, 256, device=dev)
bufs = [torch.empty(s, F, device=dev) for _ in range(2)]
t0 = time.time()
for it in range(int(sys.argv[1]) if len(sys.argv) > 1 else 400):
    pending = {}

    def post(i):
        buf = own if i == rank else bufs[i % 2]
        pending[i] = (dist.broadcast(buf, src=i, async_op=True), buf)

    post(0)
    z = torch.zeros(s, 256, device=dev)
    for i in range(world):
        if i + 1 < world:
            post(i + 1)
        work, buf = pending.pop(i)
        work.wait()
        z += buf @ W
    if it % 20 == 0:
        print(f'rank {rank} iteration {it} t={time.time() - t0:.1f}s checksum {float(z.sum()):.3e}', flush=True)
dist.barrier()
print(f'rank {rank} done in {time.time() - t0:.1f}s', flush=True)
dist.destroy_process_group()
