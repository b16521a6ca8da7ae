#!/usr/bin/env python3
"""torch.profiler view of one training epoch on one GPU (which ops own the non-SpMM time, with call sites)."""
import argparse
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'dist-gnn_amd'))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--workload', default='reddit')
    ap.add_argument('--scale', type=float, default=1.0)
    ap.add_argument('--match', default='copy')
    a = ap.parse_args()
    import bench
    import dist_train
    import dist_utils
    os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
    os.environ.setdefault('MASTER_PORT', '29534')
    os.environ.setdefault('DGNN_LOG_DIR', f'/tmp/dgnn_prof_{os.getpid()}')
    os.environ['DGNN_CACHE_INPUT'] = '0'
    env = dist_utils.DistEnv(0, 1, 'nccl', echo=False)
    args = argparse.Namespace(workload=a.workload, scale=a.scale, seed=0, model='GCN', rep=2, lean=False)
    g, _, _ = bench.build_rank_graph(args, env)
    layers = {'reddit': 2, 'products': 3, 'cora': 2}[a.workload]
    model = dist_train.build_model(g, env, 'GCN', 256, layers)
    opt = torch.optim.Adam(model.parameters(), lr=0.01)
    lf = torch.nn.CrossEntropyLoss()
    for _ in range(3):
        dist_train.train_step(model, opt, lf, g, env)
    torch.cuda.synchronize()
    from torch.profiler import profile, ProfilerActivity
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
        dist_train.train_step(model, opt, lf, g, env)
        torch.cuda.synchronize()
    print(prof.key_averages().table(sort_by='cuda_time_total', row_limit=25, max_name_column_width=60))
    evs = [e for e in prof.events() if a.match in e.name and e.device_time_total > 20]
    for e in sorted(evs, key=lambda e: -e.device_time_total)[:12]:
        print(f'{e.name}  {e.device_time_total:.0f} us  shapes?  stack:')
        for fr in (e.stack or [])[:8]:
            print('     ', fr)
    torch.distributed.destroy_process_group()


if __name__ == '__main__':
    main()
