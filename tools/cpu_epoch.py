#!/usr/bin/env python3
"""CPU/gloo epoch time of the reference's own CPU configuration, for bench.py's cpu_baseline leg.

What the reference runs without a GPU (light_dist_gnn/main.py:27-29 picks gloo + CPU tensors; models/gcn.py:13 falls back
to ``C.addmm_(A, B)``, ATen's CPU sparse addmm): P processes over gloo, the epoch region of dist_train.py:36-46 (forward,
loss, backward, Adam step). The stage loop is this package's (same arithmetic: one addmm_ per stage in source order), the
SpMM is the reference's fallback line itself, installed through the hook the CPU tests use. Never part of the product
path and never the measured value of bench.py -- a reported baseline.

    python tools/cpu_epoch.py --workload cora --nprocs 2 --epochs 3          -> one JSON line on stdout
"""
import argparse
import json
import os
import socket
import sys
import time

import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _addmm_fallback(A, B, C):
    C.addmm_(A, B)                       # light_dist_gnn/models/gcn.py:13


def _worker(rank, args, port, threads, q):
    for p in (ROOT, os.path.join(ROOT, 'dist-gnn_amd')):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), DGNN_PIPELINE='bcast', DGNN_GLOO_CUDA='0',
                      DGNN_LOG_DIR=os.path.join('/tmp', f'dgnn_cpu_epoch_{port}'))
    torch.set_num_threads(threads)
    import coo_graph
    import dist_train
    import dist_utils
    from coo_graph import synthetic
    from models import gcn
    gcn.set_spmm_backend(_addmm_fallback)
    env = dist_utils.DistEnv(rank, args.nprocs, 'gloo', echo=False)
    n, e = synthetic.SHAPES[args.workload][:2]
    n, e = int(n * args.scale), int(e * (args.scale ** 2 if args.scale < 1 else 1))
    part = synthetic.make_partition(args.workload, rank, args.nprocs, seed=0, device='cpu', num_nodes=n, num_edges=e)
    g = coo_graph.Parted_COO_Graph(args.workload, rank, args.nprocs, device='cpu', attr_dict=part)
    model = dist_train.build_model(g, env, 'GCN', args.hidden, args.layers)
    opt = torch.optim.Adam(model.parameters(), lr=0.01)
    loss_func = torch.nn.CrossEntropyLoss()
    dist_train.train_step(model, opt, loss_func, g, env)                    # warm-up (plans nothing: CPU path)
    env.barrier_all()
    t0 = time.perf_counter()
    for _ in range(args.epochs):
        dist_train.train_step(model, opt, loss_func, g, env)
    env.barrier_all()
    dt = (time.perf_counter() - t0) / args.epochs
    t = torch.tensor([dt], dtype=torch.float64)
    torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
    nnz = torch.tensor([float(g.local_num_edges)], dtype=torch.float64)
    torch.distributed.all_reduce(nnz)
    if rank == 0:
        q.put(dict(workload=args.workload, scale=args.scale, nodes=n, nnz=int(nnz.item()), P=args.nprocs, layers=args.layers,
                   hidden=args.hidden, threads_per_process=threads, epoch_s=round(float(t.item()), 5), epochs=args.epochs,
                   spmm='torch C.addmm_(A, B) on CPU (the reference fallback, gcn.py:13)', backend='gloo'))
    env.barrier_all()
    torch.distributed.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--workload', default='cora')
    ap.add_argument('--scale', type=float, default=1.0)
    ap.add_argument('--nprocs', type=int, default=1)
    ap.add_argument('--epochs', type=int, default=3)
    ap.add_argument('--layers', type=int, default=2)
    ap.add_argument('--hidden', type=int, default=256)
    ap.add_argument('--cores', type=int, default=0, help='host cores to spread over the processes (0 = this process\'s share)')
    args = ap.parse_args()
    cores = args.cores or int(os.environ.get('OMP_NUM_THREADS', 0) or 0) or len(os.sched_getaffinity(0))
    threads = max(1, cores // args.nprocs)               # torch.set_num_threads(cores // P), SURVEY 8d
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, args, port, threads, q)) for r in range(args.nprocs)]
    for p in procs:
        p.start()
    out = q.get(timeout=600)
    for p in procs:
        p.join(timeout=60)
    out['cores'] = cores
    print(json.dumps(out), flush=True)


if __name__ == '__main__':
    main()
