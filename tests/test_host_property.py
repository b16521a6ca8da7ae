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


def test_dense_plan_splits_a_block_into_tiles_and_remainder_exactly():
    """spmm_cpp.DensePlan is integer work with torch ops and runs without a GPU: tiles + remainder hold every entry of
    the block exactly once (same values), piece lengths are multiples of 8 (pad slots weigh 0), every slot belongs to
    one piece, the pieces of a cut tile own consecutive partial rows, and tiles under min_cols stay untouched."""
    import numpy as np
    import torch
    import spmm_cpp
    from coo_graph import synthetic, graph_utils
    n = 2500
    d = synthetic.make_dataset('reddit', seed=3, device='cpu', num_nodes=n, num_edges=300000, with_features=False)
    adj = graph_utils.sym_normalization(d['edge_index'], n).coalesce()
    idx, val = adj.indices(), adj.values()
    rp = torch.zeros(n + 1, dtype=torch.int32)
    rp[1:] = torch.cumsum(torch.bincount(idx[0], minlength=n), 0).int()
    A = spmm_cpp.CsrBlock(n, n, rp, idx[1].int(), val)
    for min_count, min_cols, piece in ((6, 16, 64), (3, 200, 1024)):
        D = spmm_cpp.DensePlan(A, min_count, min_cols, piece, keep_mask=True)
        assert 0.2 < D.share < 1.0 and D.density >= min_count / 32 and D.nk % 8 == 0
        rows = D.rows.long().view(-1, 32)
        desc = D.desc.long().numpy()
        dense = np.zeros((n, n))
        seen = np.zeros(D.nk, dtype=np.int64)
        kcol, kval = D.kcol.numpy(), D.kval.view(-1, 32).numpy()
        per_tile = {}
        for k0, k1, t, out in desc:
            assert k1 > k0 and (k1 - k0) % 8 == 0
            seen[k0:k1] += 1
            r = rows[t].numpy()
            ok = r >= 0
            assert (kval[k0:k1][:, ~ok] == 0).all()
            cols = kcol[k0:k1]
            assert (np.diff(cols) >= 0).all()                              # ascending (pad slots repeat the last column)
            np.add.at(dense, (np.repeat(r[ok][None, :], k1 - k0, 0), np.repeat(cols[:, None], ok.sum(), 1)), kval[k0:k1][:, ok])
            per_tile.setdefault(int(t), []).append((int(k0), int(out)))
        assert (seen == 1).all()
        R = D.rest
        ri = np.repeat(np.arange(n), np.diff(R.rowptr.numpy()))
        rest = np.zeros((n, n))
        rest[ri, R.col.numpy()] = R.val.numpy()
        assert np.array_equal(dense + rest, adj.to_dense().numpy().astype(np.float64))
        assert R.nnz + int(D.in_dense.sum()) == A.nnz
        cut = {int(c[0]): (int(c[1]), int(c[2]), int(c[3])) for c in D.cut.numpy()}
        for t, pieces in per_tile.items():
            members = rows[t][rows[t] >= 0].tolist()
            if len(pieces) == 1:
                assert pieces[0][1] == -1 and not any(r in cut for r in members)
            else:
                outs = sorted(o for _, o in sorted(pieces))
                assert outs == list(range(outs[0], outs[0] + 32 * len(pieces), 32))
                for j, r in enumerate(rows[t].tolist()):
                    if r >= 0:
                        assert cut[r] == (outs[0] + j, len(pieces), 31)
        assert int(D.touched_rows().sum()) == sum(int((rows[t] >= 0).sum()) for t in per_tile)
        assert D.npartial == 32 * sum(len(p) for p in per_tile.values() if len(p) > 1)


def test_mean_cross_entropy_spelled_out_equals_nn_cross_entropy():
    """dist_train.cross_entropy_mean replaces nn.CrossEntropyLoss() (dist_train.py:33) in the training loop: same
    loss and gradient, ragged batch sizes and one-row batches included."""
    import dist_train
    g = torch.Generator().manual_seed(3)
    for rows, classes in ((1, 7), (37, 41), (1000, 41), (513, 172)):
        x = (torch.randn(rows, classes, generator=g) * 5).requires_grad_(True)
        y = torch.randint(0, classes, (rows,), generator=g)
        want = torch.nn.CrossEntropyLoss()(x, y)
        gw, = torch.autograd.grad(want, x)
        got = dist_train.cross_entropy_mean(x, y)
        gg, = torch.autograd.grad(got, x)
        assert abs(got.item() - want.item()) <= 1e-6 * max(1.0, abs(want.item()))
        assert (gg - gw).abs().max().item() <= 1e-7


def test_gemm_tuner_and_row_pitch_helpers_without_a_gpu(monkeypatch):
    """On a CPU-only box the training drivers' GPU-side conveniences stand aside: the GEMM tuner reports off, and a CPU matrix never counts as stored at the kernels' row pitch (the pipeline then takes its
    contiguous copy, as before)."""
    import dist_utils
    import spmm_cpp
    if torch.cuda.is_available():
        return                                          # (the GPU suite covers the other side)
    monkeypatch.setenv('DGNN_TUNE_GEMM', '1')
    assert dist_utils.tune_gemms() is False and dist_utils.tune_gemms(True) is False and dist_utils.tune_gemms('library') is False
    assert dist_utils.gemm.picks() == {}              # (and no HIP call was made: there is no device)
    a, b = torch.rand(6, 5), torch.rand(5, 4)
    assert torch.equal(dist_utils.gemm.mm(a, b), torch.mm(a, b))        # CPU operands pass straight through
    x = torch.rand(5, 64)
    assert not spmm_cpp.is_row_padded(x) and spmm_cpp.padded_width(41) == 64 and spmm_cpp.padded_width(602) == 608
    try:
        spmm_cpp.row_padded(x)
        raise AssertionError('row_padded accepted a CPU tensor')
    except RuntimeError as e:
        assert 'CUDA' in str(e)
