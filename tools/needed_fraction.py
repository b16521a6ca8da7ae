#!/usr/bin/env python3
"""Share of the feature rows a rank actually needs from its peers (distinct columns of the off-diagonal blocks
over rows available), for a dataset shape and partition count: what the needed-rows form would move."""
import argparse
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'dist-gnn_amd'))
from coo_graph import synthetic, graph_utils  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--shape', default='products')
    ap.add_argument('--parts', default='2,4,8')
    ap.add_argument('--planted', type=int, default=0)
    ap.add_argument('--reorder', default=None, choices=['bfs', 'degree', 'cluster'])
    a = ap.parse_args()
    dev = torch.device('cuda:0')
    n, e = synthetic.SHAPES[a.shape][:2]
    if a.planted:                      # the shape's counts on a graph with communities of that many nodes (scrambled ids)
        keys = synthetic.planted_undirected(n, e, community=a.planted, seed=0, device=dev)
        ei = torch.stack([torch.cat([keys // n, keys % n]), torch.cat([keys % n, keys // n])])
    else:
        ei = synthetic.make_dataset(a.shape, seed=0, device=dev, with_features=False)['edge_index']
    if a.reorder:                      # what prepare_data.py --reorder gives
        from coo_graph import reorder
        order = reorder.ORDERS[a.reorder](ei, n)
        new_id = torch.empty(n, dtype=torch.long, device=dev)
        new_id[order] = torch.arange(n, device=dev)
        ei = new_id[ei]
    adj = graph_utils.sym_normalization(ei, n).coalesce()
    row, col = adj.indices()
    for P in [int(x) for x in a.parts.split(',')]:
        s = -(-n // P)
        rb, cb = row // s, col // s
        off = rb != cb
        key = torch.unique(rb[off] * (s * P) + col[off])            # (receiving rank, global column) pairs
        need = key.numel()
        print(f'{a.shape}{" planted " + str(a.planted) if a.planted else ""}{" reorder " + a.reorder if a.reorder else ""} P={P}: {need / (P * (P - 1) * s):.3f} of the rows of the peers\' blocks are referenced '
              f'({need} of {P * (P - 1) * s})', flush=True)


if __name__ == '__main__':
    main()
