"""The product's host side (coo_graph) against the reference's golden vectors: index work bit-exact."""
import os

import numpy as np
import pytest
import torch

NAMES = ['tiny', 'ragged', 'directed', 'multilabel']


def _raw(g):
    return {"edge_index": torch.from_numpy(g['edge_index']), "features": torch.from_numpy(g['raw_features']),
            "labels": torch.from_numpy(g['labels']), "train_mask": torch.from_numpy(g['train_mask']),
            "val_mask": torch.from_numpy(g['val_mask']), "test_mask": torch.from_numpy(g['test_mask']),
            "num_nodes": int(g['num_nodes']), "num_edges": g['edge_index'].shape[1], "num_classes": int(g['num_classes'])}


@pytest.mark.parametrize('name', NAMES)
def test_graph_utils_match_reference(golden_dir, name):
    from coo_graph import graph_utils
    g = np.load(os.path.join(golden_dir, f'{name}_graph.npz'))
    n = int(g['num_nodes'])
    ei = torch.from_numpy(g['edge_index'])
    assert np.array_equal(graph_utils.add_self_loops(ei, n).numpy(), g['self_loops_edge_index'])
    adj = graph_utils.sym_normalization(ei, n)
    assert adj.is_coalesced()
    assert np.array_equal(adj.indices().numpy(), g['adj_idx'])
    np.testing.assert_allclose(adj.values().numpy(), g['adj_val'], rtol=1e-6, atol=0)
    pre = graph_utils.preprocess(name, _raw(g), 'GCN')
    assert 'edge_index' not in pre
    np.testing.assert_allclose(pre['features'].numpy(), g['features'], rtol=1e-6, atol=1e-9)
    gold_adj = torch.sparse_coo_tensor(torch.from_numpy(g['adj_idx']), torch.from_numpy(g['adj_val']), (n, n)).coalesce()
    for P in (1, 2, 3, 4):
        s = (n + P - 1) // P
        for dim in (0, 1):
            parts = graph_utils.sparse_2d_split(gold_adj, s, split_dim=dim)
            for j, part in enumerate(parts):
                key = f'split_P{P}_dim{dim}_part{j}'
                assert part.is_coalesced()
                assert np.array_equal(part.indices().numpy(), g[key + '_idx'])
                assert np.array_equal(part.values().numpy(), g[key + '_val'])
                assert tuple(part.shape) == tuple(g[key + '_shape'])


@pytest.mark.parametrize('name,P', [('tiny', 1), ('tiny', 2), ('tiny', 3), ('ragged', 2), ('ragged', 3),
                                    ('directed', 2), ('multilabel', 2)])
def test_partition_and_local_view(golden_dir, name, P, tmp_path, monkeypatch):
    import coo_graph
    from coo_graph import datasets
    monkeypatch.setattr(datasets, 'data_root', str(tmp_path))
    g = np.load(os.path.join(golden_dir, f'{name}_graph.npz'))
    w = np.load(os.path.join(golden_dir, f'{name}_P{P}.npz'))
    full = coo_graph.COO_Graph(name, attr_dict=_raw(g))
    full.partition(P)                                            # through the cache files
    mem_parts = full.partition(P, save=False)
    for r in range(P):
        for pg in (coo_graph.Parted_COO_Graph(name, r, P), coo_graph.Parted_COO_Graph(name, r, P, attr_dict=mem_parts[r])):
            assert len(pg.adj_parts) == P
            for j, part in enumerate(pg.adj_parts):
                assert np.array_equal(part.indices().numpy(), w[f'r{r}_adj_part{j}_idx'])
                np.testing.assert_allclose(part.values().numpy(), w[f'r{r}_adj_part{j}_val'], rtol=1e-6)
                assert tuple(part.shape) == tuple(w[f'r{r}_adj_part{j}_shape'])
            np.testing.assert_allclose(pg.features.numpy(), w[f'r{r}_features'], rtol=1e-6, atol=1e-9)
            assert np.array_equal(pg.local_labels.numpy(), w[f'r{r}_local_labels'])
            assert np.array_equal(pg.local_train_mask.numpy(), w[f'r{r}_local_train_mask'])
            assert np.array_equal(pg.labels.numpy(), w[f'r{r}_labels_all'])
            assert np.array_equal(pg.train_mask.numpy(), w[f'r{r}_train_mask_all'])
            assert pg.local_num_nodes == pg.adj.size(0) and pg.local_num_edges == pg.adj._nnz()


@pytest.mark.parametrize('name,P', [('tiny', 2), ('ragged', 3), ('directed', 2)])
def test_part_files_carry_int32_csr_blocks(golden_dir, name, P, tmp_path, monkeypatch):
    """partition(..., with_csr=True): the stored CSR of every block equals the oracle's coo2csr of the
    reference's block (tests/golden), through the cache file and back (weights_only load)."""
    import coo_graph
    from coo_graph import datasets, parted_coo_graph
    from oracle import ref_cpu
    monkeypatch.setattr(datasets, 'data_root', str(tmp_path))
    g = np.load(os.path.join(golden_dir, f'{name}_graph.npz'))
    w = np.load(os.path.join(golden_dir, f'{name}_P{P}.npz'))
    full = coo_graph.COO_Graph(name, attr_dict=_raw(g))
    full.partition(P, with_csr=True, with_transpose=True)
    for r in range(P):
        d = parted_coo_graph.GraphCache.load_dict(parted_coo_graph.GraphCache.parted_graph_path(name, 'GCN', r, P))
        pg = coo_graph.Parted_COO_Graph(name, r, P, attr_dict=d, transposed_backward=True)
        assert pg.csr_from_file is False                      # CPU: arrays are read, nothing is attached
        for j in range(P):
            idx = w[f'r{r}_adj_part{j}_idx']
            n = int(w[f'r{r}_adj_part{j}_shape'][0])
            assert d[f'csr{j}__rowptr'].dtype == torch.int32 and d[f'csr{j}__col'].dtype == torch.int32
            assert np.array_equal(d[f'csr{j}__rowptr'].numpy(), ref_cpu.coo2csr(idx[0].astype(np.int32), n))
            assert np.array_equal(d[f'csr{j}__col'].numpy(), idx[1].astype(np.int32))
            np.testing.assert_allclose(d[f'csr{j}__val'].numpy(), w[f'r{r}_adj_part{j}_val'], rtol=1e-6)
            assert tuple(d[f'csr{j}__shape'].tolist()) == tuple(w[f'r{r}_adj_part{j}_shape'])
            t = pg.adj_parts.transposed[j].coalesce()         # the transposed blocks carry theirs too
            assert np.array_equal(d[f'csrt{j}__col'].numpy(), t.indices()[1].numpy().astype(np.int32))
            assert np.array_equal(d[f'csrt{j}__rowptr'].numpy(),
                                  ref_cpu.coo2csr(t.indices()[0].numpy().astype(np.int32), t.size(0)))


@pytest.mark.parametrize('P', [1, 3, 8])
def test_per_rank_part_generation_equals_the_full_path(P):
    """synthetic.make_partition (what a papers100M-sized run uses: no rank ever holds the whole normalised
    adjacency or feature matrix) builds the same slab, labels and masks as make_dataset -> preprocess ->
    partition_dict, bit for bit."""
    from coo_graph import synthetic, graph_utils, parted_coo_graph
    kw = dict(seed=3, num_nodes=1001, num_edges=6000)
    pre = graph_utils.preprocess('x', dict(synthetic.make_dataset('reddit', **kw)), 'GCN')
    _, parts = parted_coo_graph.partition_dict(pre, P)
    for r in range(P):
        mine = synthetic.make_partition('reddit', r, P, **kw)
        a, b = parts[r]['adj'].coalesce(), mine['adj'].coalesce()
        assert a.shape == b.shape and torch.equal(a.indices(), b.indices()) and torch.equal(a.values(), b.values())
        for k in ('labels', 'train_mask', 'val_mask', 'test_mask'):
            assert torch.equal(parts[r][k], mine[k]), k
        assert mine['features'].shape == parts[r]['features'].shape and mine['features'].dtype == torch.float32
        assert torch.equal(mine['features'], parts[r]['features'])      # drawn per row tile: the same rows at any P
        assert mine['num_nodes'] == pre['num_nodes'] and mine['num_classes'] == pre['num_classes']
        import coo_graph
        g = coo_graph.Parted_COO_Graph('x', r, P, attr_dict=mine)      # loads like any part dict
        assert len(g.adj_parts) == P and g.local_num_nodes == (1001 + P - 1) // P


def test_synthetic_dataset_schema_and_determinism():
    from coo_graph import synthetic, datasets
    a = synthetic.make_dataset('cora', seed=3)
    b = synthetic.make_dataset('cora', seed=3)
    assert set(a) == {"edge_index", "features", "labels", "train_mask", "val_mask", "test_mask",
                      "num_nodes", "num_edges", "num_classes"}
    assert torch.equal(a['edge_index'], b['edge_index']) and torch.equal(a['features'], b['features'])
    n, e = synthetic.SHAPES['cora'][:2]
    ei = a['edge_index']
    assert a['num_nodes'] == n and ei.shape == (2, 2 * e) and a['num_edges'] == 2 * e
    assert (ei[0] != ei[1]).all() and int(ei.max()) < n
    key = ei[0] * n + ei[1]
    assert key.unique().numel() == key.numel()                         # de-duplicated
    assert set(map(int, key)) == set(map(int, ei[1] * n + ei[0]))      # symmetric
    d = datasets.load_dataset('cora', num_nodes=300, num_edges=900, num_features=8)
    assert d['features'].shape == (300, 8)
    with pytest.raises(ValueError):
        datasets.load_dataset('no-such-dataset')


def test_timer_and_logger_api(tmp_path):
    from dist_utils.timer import DistTimer
    from dist_utils.logger import DistLogger

    class E:
        rank, world_size = 0, 1
    t = DistTimer(E())
    with t.timing('epoch'):
        pass
    with t.timing_cuda('spmm'):          # no GPU here: must degrade to wall clock, not raise
        pass
    t.start('x'); t.stop('x', 'sub')
    s = t.summary()
    assert 'epoch' in s and 'spmm' in s and 'x-sub' in s and t.count_dict['spmm'] == 1
    lg = DistLogger(E(), log_root=str(tmp_path), echo=False)
    lg.log('hello', 1, rank=0)
    lg.log('skipped', rank=3)
    txt = open(lg.log_fname).read()
    assert 'hello 1' in txt and 'skipped' not in txt and '[0]' in txt
