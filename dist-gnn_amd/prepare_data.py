"""Offline step: build the normalised graph and write the per-rank part files
(light_dist_gnn/prepare_data.py). Usage: python prepare_data.py --dataset cora --parts 2 4 8"""
import argparse

import coo_graph


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument('--dataset', default='cora')
    ap.add_argument('--parts', type=int, nargs='+', default=[4, 8])
    ap.add_argument('--no-cache', action='store_true')
    ap.add_argument('--no-csr', action='store_true', help='do not store the int32 CSR blocks in the part files')
    ap.add_argument('--transpose', action='store_true', help='also store the blocks of A^T (exact backward on directed graphs)')
    ap.add_argument('--reorder', default=None, choices=['bfs', 'degree', 'cluster'],
                    help='relabel the nodes for locality first (1D_CAGNET/partition_torch_ready_data.py:15-57 did this with '
                         'METIS); the result is written as dataset <name>_reorder')
    args = ap.parse_args(argv)
    if args.reorder:
        from coo_graph import datasets, reorder
        raw = datasets.load_dataset(args.dataset)
        raw['edge_index'] = raw['edge_index'].long()
        raw = reorder.reorder_dataset(raw, args.reorder)
        raw.pop('node_order')
        args.dataset = args.dataset + '_reorder'
        g = coo_graph.COO_Graph(args.dataset, attr_dict=raw)
    else:
        g = coo_graph.COO_Graph(args.dataset, full_graph_cache_enabled=not args.no_cache)
    for p in args.parts:
        g.partition(p, with_csr=not args.no_csr, with_transpose=args.transpose)
        print(coo_graph.Parted_COO_Graph(args.dataset, 0, p))
    print(g)


if __name__ == '__main__':
    main()
