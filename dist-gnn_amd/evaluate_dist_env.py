"""Broadcast / all-gather micro-benchmark over the process group (xGMI + RCCL on a GPU node, gloo on CPU).

Counterpart of light_dist_gnn/evaluate_dist_env.py:13-40: every rank takes a turn as the root of a
broadcast of an int8 buffer, for sizes 4 KB .. 2 GB with repeat counts 16000 .. 1. Added: the same volume
moved as ONE all-gather (what the P stage broadcasts of a layer pass amount to), and per-link figures
against the 153 GB/s xGMI link rate. Launch:

    torchrun --nproc-per-node 8 evaluate_dist_env.py [--backend nccl] [--max-mb 2048]
    python evaluate_dist_env.py --nprocs 2 --backend gloo --max-mb 4
"""
import argparse
import os
import time

import torch
import torch.distributed as dist

import dist_utils

XGMI_LINK_GBS = 153.0


def sizes_and_repeats(max_bytes):
    size, rep = 4 * 1024, 16000
    while size <= max_bytes:
        yield size, max(1, rep)
        size *= 8
        rep //= 8


def _sync(env):
    if env.device.type == 'cuda':
        torch.cuda.synchronize()


def evaluate(env, max_bytes):
    P = env.world_size
    rows = []
    for size, rep in sizes_and_repeats(max_bytes):
        buf = torch.zeros(size, dtype=torch.int8, device=env.device)
        gathered = torch.zeros(size * P, dtype=torch.int8, device=env.device)
        env.barrier_all(); _sync(env)
        t0 = time.perf_counter()
        for _ in range(rep):
            for src in range(P):                      # all ranks take turns (evaluate_dist_env.py:31-36)
                env.broadcast(buf, src)
        _sync(env)
        t_b = (time.perf_counter() - t0) / rep
        env.barrier_all(); _sync(env)
        t0 = time.perf_counter()
        for _ in range(rep):
            dist.all_gather_into_tensor(gathered, buf, group=env.world_group)
        _sync(env)
        t_g = (time.perf_counter() - t0) / rep
        recv = size * (P - 1)                         # bytes every rank receives per round
        rows.append((size, rep, t_b, t_g, recv / t_b / 1e9 if P > 1 else 0.0, recv / t_g / 1e9 if P > 1 else 0.0))
        env.logger.log(f'{size:>11d} B x{rep:>5d}: {P} bcasts {t_b * 1e3:9.3f} ms ({rows[-1][4]:7.2f} GB/s in per rank) | '
                       f'all_gather {t_g * 1e3:9.3f} ms ({rows[-1][5]:7.2f} GB/s in per rank, '
                       f'{rows[-1][5] / max(P - 1, 1) / XGMI_LINK_GBS:5.2f} of a {XGMI_LINK_GBS:.0f} GB/s link each)', rank=0)
    return rows


def _main(rank, args):
    os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
    os.environ.setdefault('MASTER_PORT', str(args.port))
    env = dist_utils.DistEnv(rank, args.nprocs, args.backend)
    evaluate(env, args.max_mb * 1024 * 1024)
    dist.destroy_process_group()


if __name__ == '__main__':
    ap = argparse.ArgumentParser()
    ap.add_argument('--nprocs', type=int, default=max(torch.cuda.device_count(), 1))
    ap.add_argument('--backend', default='nccl' if torch.cuda.is_available() else 'gloo')
    ap.add_argument('--max-mb', type=int, default=2048)
    ap.add_argument('--port', type=int, default=29512)
    a = ap.parse_args()
    if 'RANK' in os.environ:
        a.nprocs = int(os.environ['WORLD_SIZE'])
        _main(int(os.environ['RANK']), a)
    else:
        torch.multiprocessing.spawn(_main, (a,), a.nprocs)
