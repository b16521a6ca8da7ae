#!/usr/bin/env python3
"""Weight gradients of the first backward pass, saved per rank count, to compare N = 1 with N > 1 directly (the
loss after an Adam step amplifies rounding differences: Adam moves every weight by lr * sign(g) at step 1).
  python tools/grad_check.py --workload cora --out gpurun_out/grads_n1.pt
  python -m torch.distributed.run --nproc-per-node 3 ... tools/grad_check.py --gpus 3 --backend gloo --out gpurun_out/grads_n3.pt
  python tools/grad_check.py --compare gpurun_out/grads_n1.pt gpurun_out/grads_n3.pt"""
import argparse
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'dist-gnn_amd'))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--workload', default='cora')
    ap.add_argument('--scale', type=float, default=1.0)
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--backend', default='nccl')
    ap.add_argument('--layers', type=int, default=2)
    ap.add_argument('--hidden', type=int, default=256)
    ap.add_argument('--out', default='')
    ap.add_argument('--compare', nargs=2)
    a = ap.parse_args()
    if a.compare:
        x, y = (torch.load(p, weights_only=True) for p in a.compare)
        print('loss', x['loss'], y['loss'])
        for k in sorted(k for k in x if k.startswith('weight')):
            gx, gy = x[k].double(), y[k].double()
            print(f'{k}: max|g| {gx.abs().max():.3e}  max|dg| {(gx - gy).abs().max():.3e}  '
                  f'relative to max|g| {(gx - gy).abs().max() / gx.abs().max():.2e}')
        return
    import bench
    import dist_train
    import dist_utils
    os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
    os.environ.setdefault('MASTER_PORT', '29533')
    os.environ.setdefault('DGNN_LOG_DIR', f'/tmp/dgnn_gradcheck_{os.getpid()}')
    if a.backend == 'gloo':
        os.environ['DGNN_GLOO_CUDA'] = '1'
    rank, world = int(os.environ.get('RANK', 0)), int(os.environ.get('WORLD_SIZE', 1))
    env = dist_utils.DistEnv(rank, world, a.backend, echo=False)
    args = argparse.Namespace(workload=a.workload, scale=a.scale, seed=0, model='GCN', rep=2, lean=False)
    g, _, _ = bench.build_rank_graph(args, env)
    model = dist_train.build_model(g, env, 'GCN', a.hidden, a.layers)
    out = model(g.features)
    idx = torch.nonzero(g.local_train_mask).flatten()
    loss = torch.nn.functional.cross_entropy(out[idx], g.local_labels[idx], reduction='sum')
    t = torch.tensor([float(idx.numel())], dtype=torch.float64, device=env.device)
    torch.distributed.all_reduce(t)
    (loss / t.item()).backward()          # mean over ALL training nodes: the same objective at every N
    tot = loss.detach().double().reshape(1)
    torch.distributed.all_reduce(tot)
    if rank == 0 and a.out:
        d = {n: p.grad.detach().cpu() for n, p in model.named_parameters()}
        d['loss'] = float(tot.item() / t.item())
        torch.save(d, os.path.join(ROOT, a.out))
        print('saved', a.out, d['loss'], flush=True)
    torch.distributed.barrier()
    torch.distributed.destroy_process_group()


if __name__ == '__main__':
    main()
