"""Locality reordering (coo_graph/reorder.py, the mechanism of 1D_CAGNET/partition_torch_ready_data.py:15-57) is a
pure relabelling: the permuted input gives the permuted answer. Checked with the CPU oracle (no GPU needed)."""
import os

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _graph(seed=4, n=157, e=900, f=9):
    g = torch.Generator().manual_seed(seed)
    return {"edge_index": torch.randint(0, n, (2, e), generator=g), "features": torch.rand(n, f, generator=g),
            "labels": torch.randint(0, 5, (n,), generator=g), "train_mask": torch.rand(n, generator=g) < 0.5,
            "val_mask": torch.zeros(n, dtype=torch.bool), "test_mask": torch.zeros(n, dtype=torch.bool),
            "num_nodes": n, "num_edges": e, "num_classes": 5}


@pytest.mark.parametrize('how', ['bfs', 'degree', 'cluster'])
def test_reordering_is_a_pure_relabelling(how):
    from coo_graph import reorder, graph_utils, parted_coo_graph
    from oracle import ref_cpu
    raw = _graph()
    n = raw['num_nodes']
    new = reorder.reorder_dataset(raw, how)
    order = new['node_order']
    assert torch.equal(torch.sort(order).values, torch.arange(n))
    new_id = torch.empty(n, dtype=torch.long)
    new_id[order] = torch.arange(n)
    assert torch.equal(new['edge_index'], new_id[raw['edge_index']])          # same edges, relabelled
    for k in ('features', 'labels', 'train_mask'):
        assert torch.equal(new[k], raw[k][order])
    a = graph_utils.preprocess('x', dict(raw), 'GCN')
    b = graph_utils.preprocess('x', {k: v for k, v in new.items() if k != 'node_order'}, 'GCN')
    A, Bm = a['adj'].coalesce(), b['adj'].coalesce()
    # the normalised adjacency of the relabelled graph IS the relabelled normalised adjacency: indices exactly ...
    want_idx, want_val = ref_cpu.coalesce(new_id[A.indices()].numpy(), A.values().numpy(), (n, n))
    assert np.array_equal(Bm.indices().numpy(), want_idx)
    np.testing.assert_allclose(Bm.values().numpy(), want_val, rtol=1e-6)      # ... values to rounding (degree sums reorder)
    # ... and the aggregate of the relabelled input is the relabelled aggregate (north_star's 1e-5)
    za = torch.sparse.mm(A, a['features'])
    zb = torch.sparse.mm(Bm, b['features'])
    assert float((zb - za[order]).abs().max()) <= 1e-5 * float(za.abs().max())
    # the partition of the relabelled input: the package's split against the oracle's, bit for bit
    for P in (2, 3):
        _, parts = parted_coo_graph.partition_dict(dict(b), P)
        s = -(-n // P)
        for r, part in enumerate(parts):
            sub = part['adj'].coalesce()
            blocks = ref_cpu.sparse_2d_split(Bm.indices().numpy(), Bm.values().numpy(), (n, n), s, 0)
            assert np.array_equal(sub.indices().numpy(), blocks[r][0])
            assert np.array_equal(sub.values().numpy(), blocks[r][1])


def test_bfs_order_puts_neighbours_close():
    """A ring with scrambled ids: after the breadth-first relabelling every edge joins ids at most 2 apart."""
    from coo_graph import reorder
    n = 101
    perm = torch.randperm(n, generator=torch.Generator().manual_seed(1))
    ring = torch.stack([torch.arange(n), (torch.arange(n) + 1) % n])
    raw = {"edge_index": perm[ring], "features": torch.zeros(n, 1), "labels": torch.zeros(n, dtype=torch.long),
           "train_mask": torch.ones(n, dtype=torch.bool), "val_mask": torch.zeros(n, dtype=torch.bool),
           "test_mask": torch.zeros(n, dtype=torch.bool), "num_nodes": n, "num_edges": n, "num_classes": 1}
    assert int((raw['edge_index'][0] - raw['edge_index'][1]).abs().max()) > 50
    new = reorder.reorder_dataset(raw, 'bfs')
    assert int((new['edge_index'][0] - new['edge_index'][1]).abs().max()) <= 2
    with pytest.raises(ValueError):
        reorder.relabel(raw, torch.zeros(n, dtype=torch.long))


def test_cluster_order_recovers_planted_communities():
    """Label-propagation clustering on a graph with communities behind scrambled ids (synthetic.planted_undirected):
    after the relabelling most edges join nodes whose ids are less than a community apart; the breadth-first numbering,
    thrown off by the 10 % long-range edges, does not get there."""
    from coo_graph import reorder, synthetic
    n, c = 12000, 400
    keys = synthetic.planted_undirected(n, 120000, community=c, seed=2)
    lo, hi = keys // n, keys % n
    assert keys.numel() >= 0.95 * 120000 and bool((lo < hi).all())
    ei = torch.stack([torch.cat([lo, hi]), torch.cat([hi, lo])])

    def near(order):
        new = torch.empty(n, dtype=torch.long)
        new[order] = torch.arange(n)
        return float(((new[ei[0]] - new[ei[1]]).abs() < c).float().mean())
    assert near(torch.arange(n)) < 0.2                                   # scrambled ids: no locality to begin with
    got = near(reorder.cluster_order(ei, n))
    assert got > 0.8, got
    assert near(reorder.bfs_order(ei, n)) < got
    assert torch.equal(reorder.cluster_order(ei, n), reorder.cluster_order(ei, n))      # deterministic


def _cpu_block(ei, n):
    import spmm_cpp
    from coo_graph import graph_utils
    adj = graph_utils.sym_normalization(ei, n).coalesce()
    idx, val = adj.indices(), adj.values()
    rp = torch.zeros(n + 1, dtype=torch.int32)
    rp[1:] = torch.cumsum(torch.bincount(idx[0], minlength=n), 0).int()
    return spmm_cpp.CsrBlock(n, n, rp, idx[1].int(), val)


def test_locality_share_tells_a_clustered_graph_from_a_scrambled_one():
    """spmm_cpp.locality_share (what decides the row-bucketed work order): high after the cluster order on a graph with
    communities, low on the same graph with scrambled ids and on R-MAT."""
    import spmm_cpp
    from coo_graph import reorder, synthetic
    n = 40000
    keys = synthetic.planted_undirected(n, 400000, community=2048, seed=1)
    lo, hi = keys // n, keys % n
    ei = torch.stack([torch.cat([lo, hi]), torch.cat([hi, lo])])
    assert spmm_cpp.locality_share(_cpu_block(ei, n)) < 0.3
    order = reorder.cluster_order(ei, n)
    new = torch.empty(n, dtype=torch.long)
    new[order] = torch.arange(n)
    assert spmm_cpp.locality_share(_cpu_block(new[ei], n)) > 0.6
    d = synthetic.make_dataset('reddit', seed=0, device='cpu', num_nodes=n, num_edges=1500000, with_features=False)
    assert spmm_cpp.locality_share(_cpu_block(d['edge_index'], n)) < 0.3


def test_auto_reorder_keeps_the_cluster_order_only_where_it_pays():
    """prepare_data.py's default (--reorder auto): a scrambled graph WITH communities comes back in the cluster order and
    passes the plans' own locality test; R-MAT (nothing to recover) and an already local graph come back unchanged."""
    from coo_graph import reorder, synthetic
    n = 40000
    keys = synthetic.planted_undirected(n, 400000, community=2048, seed=1)
    ei = torch.stack([keys // n, keys % n])
    d = dict(edge_index=ei, num_nodes=n, labels=torch.arange(n))
    out, rep = reorder.auto_reorder(d)
    assert rep['order'] == 'cluster' and rep['before'] < 0.3 and rep['after'] >= 0.5
    assert torch.equal(out['labels'], out['node_order'])                # per-node data travelled with the nodes
    assert abs(reorder.locality_share(out['edge_index'], n) - rep['after']) < 1e-12
    again, rep2 = reorder.auto_reorder({k: v for k, v in out.items() if k != 'node_order'})
    assert rep2['order'] == 'none' and again['edge_index'] is out['edge_index']
    r = synthetic.make_dataset('reddit', seed=0, device='cpu', num_nodes=n, num_edges=600000, with_features=False)
    same, rep3 = reorder.auto_reorder(dict(edge_index=r['edge_index'], num_nodes=n))
    assert rep3['order'] == 'none' and rep3['after'] < 0.5 and 'node_order' not in same
