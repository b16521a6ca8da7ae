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
    ap.add_argument('--reorder', default='auto', choices=['auto', 'none', 'bfs', 'degree', 'cluster'],
                    help='relabel the nodes for locality first (1D_CAGNET/partition_torch_ready_data.py:15-57 did this with '
                         'METIS); the result is written as dataset <name>_reorder. auto (default): the cluster order, kept '
                         'only if the relabelled graph passes the SpMM plans\' own locality test (share >= 0.5, so the '
                         'row-bucket order engages) -- a graph without clusters (R-MAT) is written as it is, under its own '
                         'name; none: the reference\'s prepare_data.py')
    args = ap.parse_args(argv)
    raw = None
    if args.reorder != 'none':
        import torch
        from coo_graph import datasets, reorder
        raw = datasets.load_dataset(args.dataset)
        raw['edge_index'] = raw['edge_index'].long()
        if args.reorder == 'auto':
            dev = 'cuda' if torch.cuda.is_available() else 'cpu'         # label propagation sorts the edge list 8 times
            work = dict(raw, edge_index=raw['edge_index'].to(dev))
            picked, report = reorder.auto_reorder(work)
            print('reorder auto: locality share %.3f -> %.3f with the cluster order: %s (%s)'
                  % (report['before'], report['after'], report['order'], report['why']))
            raw = reorder.relabel(raw, picked['node_order'].cpu()) if report['order'] != 'none' else None
        else:
            raw = reorder.reorder_dataset(raw, args.reorder)
    if raw is not None:
        raw.pop('node_order')
        args.dataset = args.dataset + '_reorder'
        print(f'writing the relabelled graph as dataset {args.dataset} (pass that name to main.py)')
        g = coo_graph.COO_Graph(args.dataset, attr_dict=raw)
    else:
        g = coo_graph.COO_Graph(args.dataset, full_graph_cache_enabled=not args.no_cache)
    for p in args.parts:
        g.partition(p, with_csr=not args.no_csr, with_transpose=args.transpose)
        print(coo_graph.Parted_COO_Graph(args.dataset, 0, p))
    print(g)


if __name__ == '__main__':
    main()
