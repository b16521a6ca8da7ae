"""Launcher: light_dist_gnn/main.py (--nprocs --epoch --backend --dataset; one process per GPU).

Two ways in, both ending in ``dist_train.main(env, args)``:
  python main.py --nprocs 8 --dataset reddit          spawns the ranks itself (as the reference)
  torchrun --nproc-per-node 8 main.py --dataset ...   ranks come from RANK/WORLD_SIZE
Rendezvous is 127.0.0.1 (single node, like the reference, main.py:10-12).
"""
import argparse
import os

import torch

import dist_train
import dist_utils


def process_wrapper(rank, args, func):
    os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
    os.environ.setdefault('MASTER_PORT', str(args.port))
    os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    os.environ['LOCAL_RANK'] = os.environ.get('LOCAL_RANK', str(rank))
    env = dist_utils.DistEnv(rank, args.nprocs, args.backend)
    try:
        func(env, args)
    finally:
        if torch.distributed.is_initialized():
            torch.distributed.destroy_process_group()


def parse_args(argv=None):
    n_gpu = torch.cuda.device_count()
    ap = argparse.ArgumentParser()
    ap.add_argument('--nprocs', type=int, default=max(n_gpu, 1))
    ap.add_argument('--epoch', type=int, default=20)
    ap.add_argument('--backend', type=str, default='nccl' if n_gpu >= 1 else 'gloo')
    ap.add_argument('--dataset', type=str, default='reddit')
    ap.add_argument('--model', type=str, default='GCN', choices=['GCN', 'CachedGCN'])
    ap.add_argument('--hidden', type=int, default=256)
    ap.add_argument('--layers', type=int, default=3)
    ap.add_argument('--port', type=int, default=29500)
    return ap.parse_args(argv)


if __name__ == '__main__':
    args = parse_args()
    if 'RANK' in os.environ and 'WORLD_SIZE' in os.environ:      # launched by torchrun
        args.nprocs = int(os.environ['WORLD_SIZE'])
        if args.nprocs > 1 and os.environ.get('DGNN_SUPERVISED') != '1' and os.environ.get('DGNN_SUPERVISE', '1') == '1':
            # stay off the GPU and run the real rank as a child: an exchange form that hangs in 'auto' then costs one
            # restart of fresh rank processes without it, not the job (dist_utils/supervise.py; DGNN_SUPERVISE=0: direct)
            import sys
            raise SystemExit(dist_utils.supervise.run([sys.executable, os.path.abspath(__file__)] + sys.argv[1:]))
        process_wrapper(int(os.environ['RANK']), args, dist_train.main)
    else:
        torch.multiprocessing.spawn(process_wrapper, (args, dist_train.main), args.nprocs)
