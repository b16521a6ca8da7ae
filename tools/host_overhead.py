#!/usr/bin/env python3
"""Host-side cost of one pipeline stage (GPU only): how many microseconds of Python/ctypes/torch a stage
costs when the kernels themselves are tiny. At N = 8 an epoch issues 32 stages per layer pass pair, so this
bounds the epoch from below when blocks are small."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'dist-gnn_amd'))
import spmm_cpp  # noqa: E402
from models import gcn  # noqa: E402

dev = torch.device('cuda:0')
n = 2000
idx = torch.randint(0, n, (2, 20000))
A = torch.sparse_coo_tensor(idx, torch.rand(20000), (n, n)).coalesce().to(dev)
for F in (602, 256, 41):
    H = torch.rand(n, F, device=dev)
    Z = torch.zeros(n, F, device=dev)
    Hp = spmm_cpp.pad_rows(H)
    blk = gcn._csr_block(A)
    for name, fn in (('gcn.spmm (pad + plan)', lambda: gcn.spmm(A, H, Z)),
                     ('gcn.spmm (padded view)', lambda: gcn.spmm(A, Hp[:, :F], Z)),
                     ('spmm_cpp.spmm_plan', lambda: spmm_cpp.spmm_plan(blk, Hp, Z)),
                     ('torch.mm 8x8 (reference point)', lambda: torch.mm(Z[:8, :8], Z[:8, :8]))):
        for _ in range(20):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(2000):
            fn()
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        print(f'F={F:4d} {name:32s} host {1e6 * (t1 - t0) / 2000:7.1f} us/call   (drain {1e3 * (t2 - t1):.1f} ms)', flush=True)

# ---- the whole stage loop (broadcast post + wait + timers + SpMM) on one rank over RCCL, 8 stages
import socket
import dist_utils
with socket.socket() as sk:
    sk.bind(('127.0.0.1', 0))
    port = sk.getsockname()[1]
os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), DGNN_LOG_DIR='/tmp/dgnn_ho_logs')
env = dist_utils.DistEnv(0, 1, 'nccl', echo=False)
stages = [(0, 0)] * 8
for F in (602, 256, 41):
    H = torch.rand(n, F, device=dev)
    pipe = gcn.StagePipeline(mode='bcast')
    for name, fn in (('bcast form, 8 stages', lambda: pipe._run_bcast([A], H, stages, env.world_group, torch.zeros(n, F, device=dev), None, None)),
                     ('allgather form (1 collective + 8 SpMMs)', lambda: pipe._run_allgather([A], H, [(0, 0)], env.world_group, torch.zeros(n, F, device=dev), None))):
        for _ in range(10):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(200):
            fn()
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        print(f'F={F:4d} {name:42s} host {1e6 * (t1 - t0) / 200:7.1f} us/aggregation', flush=True)
torch.distributed.destroy_process_group()
