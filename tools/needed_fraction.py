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
    a = ap.parse_args()
    dev = torch.device('cuda:0')
    n, e = synthetic.SHAPES[a.shape][:2]
    d = synthetic.make_dataset(a.shape, seed=0, device=dev, with_features=False)
    adj = graph_utils.sym_normalization(d['edge_index'], n).coalesce()
    row, col = adj.indices()
    for P in [int(x) for x in a.parts.split(',')]:
        s = -(-n // P)
        rb, cb = row // s, col // s
        off = rb != cb
        key = torch.unique(rb[off] * (s * P) + col[off])            # (receiving rank, global column) pairs
        need = key.numel()
        print(f'{a.shape} P={P}: {need / (P * (P - 1) * s):.3f} of the rows of the peers\' blocks are referenced '
              f'({need} of {P * (P - 1) * s})', flush=True)


if __name__ == '__main__':
    main()
