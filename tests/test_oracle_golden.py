"""The CPU oracle (oracle/) against the golden vectors captured from the reference's own Python
(tests/golden/make_golden.py). Integer/index results must be bit-exact; fp32 values within 1e-6."""
import glob
import os

import numpy as np
import pytest

from oracle import ref_cpu

NAMES = ['tiny', 'ragged', 'directed', 'multilabel']


def _load(golden_dir, fname):
    return np.load(os.path.join(golden_dir, fname), allow_pickle=False)


@pytest.mark.parametrize('name', NAMES)
def test_normalisation_and_split(golden_dir, name):
    g = _load(golden_dir, f'{name}_graph.npz')
    n = int(g['num_nodes'])
    sl = ref_cpu.add_self_loops(g['edge_index'], n)
    assert np.array_equal(sl, g['self_loops_edge_index'])
    idx, val = ref_cpu.sym_normalization(g['edge_index'], n)
    assert np.array_equal(idx, g['adj_idx'])                       # index work: bit-exact
    np.testing.assert_allclose(val, g['adj_val'], rtol=1e-6, atol=0)
    np.testing.assert_allclose(ref_cpu.normalize_features(g['raw_features']), g['features'], rtol=1e-6, atol=1e-9)
    for P in (1, 2, 3, 4):
        s = (n + P - 1) // P
        for dim in (0, 1):
            parts = ref_cpu.sparse_2d_split(g['adj_idx'], g['adj_val'], (n, n), s, dim)
            for j, (pi, pv, shp) in enumerate(parts):
                key = f'split_P{P}_dim{dim}_part{j}'
                assert np.array_equal(pi, g[key + '_idx'])
                assert np.array_equal(pv, g[key + '_val'])
                assert tuple(g[key + '_shape']) == tuple(shp)
            assert f'split_P{P}_dim{dim}_part{len(parts)}_idx' not in g.files


def _golden_worlds(golden_dir):
    out = []
    for f in sorted(glob.glob(os.path.join(golden_dir, '*_P*.npz'))):
        base = os.path.basename(f)[:-4]
        name, p = base.rsplit('_P', 1)
        out.append((name, int(p)))
    return out


def _world_params():
    here = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')
    return _golden_worlds(here)


@pytest.mark.parametrize('name,P', _world_params())
def test_partition_and_layer(golden_dir, name, P):
    g = _load(golden_dir, f'{name}_graph.npz')
    w = _load(golden_dir, f'{name}_P{P}.npz')
    n = int(g['num_nodes'])
    masks = {k: g[k] for k in ('train_mask', 'val_mask', 'test_mask')}
    part = ref_cpu.partition(g['adj_idx'], g['adj_val'], n, g['features'], g['labels'], masks, P)
    views = [ref_cpu.local_view(part, r) for r in range(P)]
    for r in range(P):
        for j, (pi, pv, shp) in enumerate(views[r]['adj_parts']):
            assert np.array_equal(pi, w[f'r{r}_adj_part{j}_idx'])
            np.testing.assert_allclose(pv, w[f'r{r}_adj_part{j}_val'], rtol=1e-6)
            assert tuple(w[f'r{r}_adj_part{j}_shape']) == tuple(shp)
        np.testing.assert_allclose(views[r]['features'], w[f'r{r}_features'], rtol=1e-6, atol=1e-9)
        assert np.array_equal(views[r]['local_labels'], w[f'r{r}_local_labels'])
        assert np.array_equal(views[r]['local_train_mask'], w[f'r{r}_local_train_mask'])
    # hot loop, driven by the golden blocks/features so the comparison isolates the SpMM chain
    feats = [w[f'r{r}_features'] for r in range(P)]
    gouts = [w[f'r{r}_layer_grad_out'] for r in range(P)]
    gw_sum = 0
    for r in range(P):
        blocks = [(w[f'r{r}_adj_part{j}_idx'], w[f'r{r}_adj_part{j}_val'], tuple(w[f'r{r}_adj_part{j}_shape']))
                  for j in range(P)]
        W = w[f'r{r}_layer_W']
        z, out = ref_cpu.layer_forward(blocks, feats, W)
        assert np.array_equal(z, w[f'r{r}_layer_z'])               # same chain as ATen addmm_: bit-exact
        np.testing.assert_allclose(out, w[f'r{r}_layer_out'], rtol=1e-5, atol=1e-6)
        gf, gw = ref_cpu.layer_backward(blocks, gouts, feats[r], W)
        np.testing.assert_allclose(gf, w[f'r{r}_layer_grad_features'], rtol=1e-5, atol=1e-6)
        gw_sum = gw_sum + gw
    np.testing.assert_allclose(gw_sum, w['r0_layer_grad_weight'], rtol=1e-5, atol=1e-5)   # after all-reduce


@pytest.mark.parametrize('name', ['tiny', 'ragged'])
def test_partition_independence(golden_dir, name):
    """cat_i(sum_j A_ij H_j) == A_hat @ H on the unpartitioned graph, for every P."""
    g = _load(golden_dir, f'{name}_graph.npz')
    n = int(g['num_nodes'])
    for P in (1, 2, 3):
        w = _load(golden_dir, f'{name}_P{P}.npz')
        z = np.concatenate([w[f'r{r}_layer_z'] for r in range(P)])[:n]
        np.testing.assert_allclose(z, g['AX_f64'], rtol=1e-5, atol=1e-7)


def test_coo2csr_and_edge_cases():
    assert np.array_equal(ref_cpu.coo2csr(np.array([], dtype=np.int32), 4), [0, 0, 0, 0, 0])
    assert np.array_equal(ref_cpu.coo2csr(np.array([1, 1, 3], dtype=np.int32), 5), [0, 0, 2, 2, 3, 3])
    C = np.ones((2, 3), dtype=np.float32)
    ref_cpu.spmm_coo(np.array([], dtype=np.int32), np.array([], dtype=np.int32), np.array([], dtype=np.float32),
                     2, 2, np.ones((2, 3), dtype=np.float32), C, 1.0, 0.5)
    assert (C == 0.5).all()
    C[:] = np.nan
    ref_cpu.spmm_coo(np.array([0], dtype=np.int32), np.array([1], dtype=np.int32), np.array([2.0], dtype=np.float32),
                     2, 2, np.arange(6, dtype=np.float32).reshape(2, 3), C, 1.0, 0.0)
    assert np.array_equal(C, [[6, 8, 10], [0, 0, 0]])


def test_threaded_oracle_is_bit_identical():
    """The multi-core CPU baseline (rows dealt to OpenMP threads) keeps the sequential chain per row."""
    from oracle import ref_cpu
    rng = np.random.default_rng(5)
    n, m, F = 700, 500, 37
    deg = rng.integers(0, 40, n)
    rowptr = np.concatenate([[0], np.cumsum(deg)]).astype(np.int32)
    col = np.concatenate([np.sort(rng.choice(m, d, replace=False)) for d in deg]).astype(np.int32)
    val = rng.random(rowptr[-1], dtype=np.float32)
    B = rng.random((m, F), dtype=np.float32)
    C0 = rng.random((n, F), dtype=np.float32)
    one = ref_cpu.spmm_csr(rowptr, col, val, B, C0.copy(), 0.5, 2.0)
    many = ref_cpu.spmm_csr(rowptr, col, val, B, C0.copy(), 0.5, 2.0, threads=4)
    assert np.array_equal(one, many)
    part = ref_cpu.spmm_csr(rowptr, col, val, B, C0.copy(), 0.5, 2.0, r0=100, r1=300, threads=3)
    assert np.array_equal(part[100:300], one[100:300]) and np.array_equal(part[:100], C0[:100])
