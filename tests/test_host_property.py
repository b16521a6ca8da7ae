"""Randomised parity of the host-side index work against the oracle's restatement of the reference (hypothesis,
derandomised): normalisation, the one-pass 2-D split, partition + local view with padding, the stored CSR. Index
work is bit-exact; values within 1e-6 (graph_utils.py:24-40 computes them in fp32)."""
import numpy as np
import torch
from hypothesis import HealthCheck, assume, given, settings, strategies as st

from oracle import ref_cpu


@st.composite
def _graphs(draw):
    n = draw(st.integers(1, 90))
    e = draw(st.integers(0, 400))
    seed = draw(st.integers(0, 2 ** 20))
    rng = np.random.default_rng(seed)
    ei = rng.integers(0, n, (2, e)).astype(np.int64)          # duplicates and self loops included on purpose
    if draw(st.booleans()) and e:                             # undirected variant
        ei = np.concatenate([ei, ei[::-1]], axis=1)
    return dict(n=n, ei=ei, P=draw(st.integers(1, 9)), F=draw(st.integers(1, 9)), seed=seed)


@settings(max_examples=150, deadline=None, derandomize=True, suppress_health_check=list(HealthCheck))
@given(_graphs())
def test_normalise_split_partition_match_the_oracle(c):
    import coo_graph
    from coo_graph import graph_utils, parted_coo_graph
    n, ei, P = c['n'], c['ei'], c['P']
    rng = np.random.default_rng(c['seed'] + 1)
    # normalisation
    want_idx, want_val = ref_cpu.sym_normalization(ei, n)
    adj = graph_utils.sym_normalization(torch.from_numpy(ei), n).coalesce()
    assert np.array_equal(adj.indices().numpy(), want_idx)
    np.testing.assert_allclose(adj.values().numpy(), want_val, rtol=1e-6)
    # the 2-D split along either dimension, any slab size
    for dim in (0, 1):
        size = int(rng.integers(1, n + 2))
        got = graph_utils.sparse_2d_split(adj, size, dim)
        want = ref_cpu.sparse_2d_split(want_idx, adj.values().numpy(), (n, n), size, dim)
        assert len(got) == len(want)
        for g, (wi, wv, shp) in zip(got, want):
            g = g.coalesce()
            assert tuple(g.shape) == tuple(shp) and np.array_equal(g.indices().numpy(), wi)
            assert np.array_equal(g.values().numpy(), wv)          # values only move: same bits
    # partition + local view (padding when P does not divide n), CSR next to the blocks
    feats = rng.random((n, c['F']), dtype=np.float32)
    labels = rng.integers(0, 3, n).astype(np.int64)
    masks = {k: rng.random(n) < 0.5 for k in ('train_mask', 'val_mask', 'test_mask')}
    raw = {'adj': adj, 'features': torch.from_numpy(feats), 'labels': torch.from_numpy(labels), 'num_nodes': n,
           'num_edges': ei.shape[1], 'num_classes': 3, **{k: torch.from_numpy(v) for k, v in masks.items()}}
    want_part = ref_cpu.partition(want_idx, adj.values().numpy(), n, feats, labels, masks, P)
    # fewer row slabs than ranks (e.g. 9 nodes on 4 ranks: slabs of 3): the reference hangs the whole padding on its
    # last slab and has no part for the last rank (parted_coo_graph.py:67-86) -- nothing to be equal to
    assume(len(want_part['adj']) == P)
    _, parts = parted_coo_graph.partition_dict(raw, P, with_csr=True)
    s = want_part['split_size']
    for r in range(P):
        pg = coo_graph.Parted_COO_Graph('prop', r, P, attr_dict=parts[r])
        view = ref_cpu.local_view(want_part, r)
        assert pg.local_num_nodes == s
        for j, (wi, wv, shp) in enumerate(view['adj_parts']):
            blk = pg.adj_parts[j].coalesce()
            assert np.array_equal(blk.indices().numpy(), wi) and np.array_equal(blk.values().numpy(), wv)
            assert np.array_equal(parts[r][f'csr{j}__rowptr'].numpy(), ref_cpu.coo2csr(wi[0].astype(np.int32), blk.size(0)))
            assert np.array_equal(parts[r][f'csr{j}__col'].numpy(), wi[1].astype(np.int32))
        assert np.array_equal(pg.features.numpy(), view['features'])
        assert np.array_equal(pg.local_labels.numpy(), view['local_labels'])
        assert np.array_equal(pg.local_train_mask.numpy(), view['local_train_mask'])
