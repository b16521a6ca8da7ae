"""How much would bundling R rows (sharing gathered B rows) cut the gather count? (analysis only)"""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'dist-gnn_amd'))
from coo_graph import synthetic, graph_utils
dev = torch.device('cuda:0')
shape = sys.argv[1] if len(sys.argv) > 1 else 'reddit'
n, e = synthetic.SHAPES[shape][:2]
d = synthetic.make_dataset(shape, seed=0, device=dev, with_features=False)
adj = graph_utils.sym_normalization(d['edge_index'], n)
row, col = adj.indices()
nnz = row.numel()
deg = torch.bincount(row, minlength=n)
for nstage in (1, 4):
    cbw = -(-n // nstage)
    cb = col // cbw
    # per (row, stage) length
    key = row * nstage + cb
    seglen = torch.bincount(key, minlength=n * nstage).view(n, nstage)
    for R in (2, 4, 8, 16):
        tot_union, tot_sum, tot_good_union, tot_good_sum = 0, 0, 0, 0
        for s in range(nstage):
            order = torch.argsort(seglen[:, s], descending=True, stable=True)
            rank_of_row = torch.empty_like(order); rank_of_row[order] = torch.arange(n, device=dev)
            m = cb == s
            b = rank_of_row[row[m]] // R                 # bundle id of each nnz
            k = b * n + col[m]
            uniq_b = torch.unique(k) // n
            nb = (n + R - 1) // R
            usz = torch.bincount(uniq_b, minlength=nb)
            ssz = torch.bincount(b, minlength=nb)
            good = usz.float() <= 0.7 * ssz.float()
            tot_union += int(usz.sum()); tot_sum += int(ssz.sum())
            tot_good_union += int(usz[good].sum()); tot_good_sum += int(ssz[good].sum())
        gathers = tot_sum - tot_good_sum + tot_good_union
        print(f'nstage={nstage} R={R}: all-bundled gathers {tot_union/tot_sum:.3f} of nnz; selective (union<=0.7 sum): '
              f'{tot_good_sum/tot_sum:.3f} of nnz bundled -> {tot_good_union/max(tot_good_sum,1):.3f} of theirs; total gathers {gathers/tot_sum:.3f} of nnz', flush=True)
