"""oracle/ref_cpu.py -- TEST INFRASTRUCTURE ONLY (the CPU oracle, host-side half).

numpy restatement of the reference's algorithm for the GCN aggregation path; the SpMM inner
loop is the C restatement in oracle/spmm_ref.c (loaded through ctypes). Only tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module; the product
(dist-gnn_amd/) never does.

Every function cites the reference lines it restates (paths relative to the reference repo,
light_dist_gnn/ unless stated). Parity of this oracle is pinned by tests/golden/*.npz, vectors
produced by running the reference's own Python (tests/golden/make_golden.py).
"""
import ctypes
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, '_build', 'liboracle.so')
_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            raise RuntimeError(f'{_SO} missing: run `make -C oracle` (or __graft_entry__.build())')
        _lib = ctypes.CDLL(_SO)
    return _lib


def _ptr(a):
    return a.ctypes.data_as(ctypes.c_void_p)


def _f32c(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _i32c(a):
    return np.ascontiguousarray(a, dtype=np.int32)


# --------------------------------------------------------------------------- native op
def coo2csr(row, n):
    """cusparseXcoo2csr (spmm_cpp/spmm_original.cpp:141-146): sorted row ids -> rowptr[n+1]."""
    row = _i32c(row)
    rowptr = np.empty(n + 1, dtype=np.int32)
    lib().oracle_coo2csr_i32(_ptr(row), ctypes.c_int64(row.size), ctypes.c_int32(n), _ptr(rowptr))
    return rowptr


def spmm_coo(row, col, val, n, m, B, C, alpha=1.0, beta=1.0):
    """In-place C = alpha*A@B + beta*C (spmm_cpp/spmm.cpp:17-33; fallback models/gcn.py:13)."""
    row, col, val = _i32c(row), _i32c(col), _f32c(val)
    assert B.dtype == np.float32 and C.dtype == np.float32 and B.flags.c_contiguous and C.flags.c_contiguous
    F = C.shape[1]
    assert B.shape == (m, F) and C.shape == (n, F)
    lib().oracle_spmm_coo_f32(_ptr(row), _ptr(col), _ptr(val), ctypes.c_int64(val.size),
                              ctypes.c_int32(n), ctypes.c_int32(m), _ptr(B), ctypes.c_int64(F),
                              _ptr(C), ctypes.c_int64(F), ctypes.c_int32(F),
                              ctypes.c_float(alpha), ctypes.c_float(beta))
    return C


def spmm_csr(rowptr, col, val, B, C, alpha=1.0, beta=1.0, r0=0, r1=None, threads=1):
    """C = alpha*A@B + beta*C over rows [r0, r1). threads > 1: rows dealt to OpenMP threads, same bits."""
    rowptr, col, val = _i32c(rowptr), _i32c(col), _f32c(val)
    assert B.dtype == np.float32 and C.dtype == np.float32 and B.flags.c_contiguous and C.flags.c_contiguous
    F = C.shape[1]
    r1 = C.shape[0] if r1 is None else r1
    if threads > 1:
        lib().oracle_spmm_csr_f32_mt(_ptr(rowptr), _ptr(col), _ptr(val), ctypes.c_int32(r0), ctypes.c_int32(r1),
                                     _ptr(B), ctypes.c_int64(F), _ptr(C), ctypes.c_int64(F), ctypes.c_int32(F),
                                     ctypes.c_float(alpha), ctypes.c_float(beta), ctypes.c_int32(threads))
        return C
    lib().oracle_spmm_csr_f32(_ptr(rowptr), _ptr(col), _ptr(val), ctypes.c_int32(r0), ctypes.c_int32(r1),
                              _ptr(B), ctypes.c_int64(F), _ptr(C), ctypes.c_int64(F), ctypes.c_int32(F),
                              ctypes.c_float(alpha), ctypes.c_float(beta))
    return C


def spmm_csr_f64(rowptr, col, val, B, Cin, alpha=1.0, beta=1.0):
    """fp64 value of the same sum and the sum of |terms| (error yardstick for the tests)."""
    rowptr, col, val = _i32c(rowptr), _i32c(col), _f32c(val)
    B, Cin = _f32c(B), _f32c(Cin)
    n, F = Cin.shape
    out = np.empty((n, F), dtype=np.float64)
    ab = np.empty((n, F), dtype=np.float64)
    lib().oracle_spmm_csr_f64(_ptr(rowptr), _ptr(col), _ptr(val), ctypes.c_int32(n), _ptr(B), ctypes.c_int64(F),
                              _ptr(Cin), ctypes.c_int64(F), ctypes.c_int32(F), ctypes.c_float(alpha),
                              ctypes.c_float(beta), _ptr(out), _ptr(ab))
    return out, ab


# --------------------------------------------------------------------------- sparse helpers
def coalesce(idx, val, shape):
    """torch.sparse_coo_tensor(...).coalesce(): row-major sort, duplicates summed (graph_utils.py:48)."""
    idx = np.asarray(idx, dtype=np.int64).reshape(2, -1)
    val = np.asarray(val)
    key = idx[0] * np.int64(shape[1]) + idx[1]
    order = np.argsort(key, kind='stable')
    key, val = key[order], val[order]
    uniq, start = np.unique(key, return_index=True)
    summed = np.add.reduceat(val, start) if val.size else val
    return np.stack([uniq // shape[1], uniq % shape[1]]).astype(np.int64), summed.astype(val.dtype)


def add_self_loops(edge_index, num_nodes):
    """graph_utils.py:18-21: drop existing self loops, append (i, i) for every node."""
    edge_index = np.asarray(edge_index)
    keep = edge_index[0] != edge_index[1]
    allv = np.arange(num_nodes, dtype=edge_index.dtype)
    return np.concatenate([edge_index[:, keep], np.stack([allv, allv])], axis=1)


def sym_normalization(edge_index, num_nodes):
    """graph_utils.py:24-40: A = coalesce(ones); d = colsum(A)^-1/2; (D A) D; coalesce."""
    ei = add_self_loops(edge_index, num_nodes)
    idx, a = coalesce(ei, np.ones(ei.shape[1], dtype=np.float32), (num_nodes, num_nodes))
    deg = np.zeros(num_nodes, dtype=np.float32)
    np.add.at(deg, idx[1], a)                                  # torch.sparse.sum(A, 0): column sums
    d = np.power(deg, np.float32(-0.5), dtype=np.float32)
    val = (d[idx[0]] * a) * d[idx[1]]                          # sparse.mm(sparse.mm(D, A), D)
    return idx, val.astype(np.float32)


def normalize_features(x):
    """graph_utils.py:8: features / rowsum.clamp(min=1)."""
    x = np.asarray(x, dtype=np.float32)
    return x / np.maximum(x.sum(axis=1, keepdims=True, dtype=np.float32), np.float32(1.0))


def sparse_2d_split(idx, val, shape, split_size, split_dim=0):
    """graph_utils.py:43-55: mask one index dim into [lower, upper) slabs, re-base it, coalesce."""
    size = shape[split_dim]
    seps = list(range(0, size, split_size)) + [size]
    parts = []
    for lower, upper in zip(seps[:-1], seps[1:]):
        mask = (idx[split_dim] < upper) & (idx[split_dim] >= lower)
        sub = idx[:, mask].copy()
        sub[split_dim] -= lower
        shp = (upper - lower, shape[1]) if split_dim == 0 else (shape[0], upper - lower)
        pi, pv = coalesce(sub, val[mask], shp)
        parts.append((pi, pv, shp))
    return parts


def partition(idx, val, num_nodes, features, labels, masks, num_parts):
    """COO_Graph.partition (parted_coo_graph.py:63-93): s = ceil(N/P); row-split adj; zero-pad the
    last feature block, labels and masks; every adj block is declared (s, s*P)."""
    s = (num_nodes + num_parts - 1) // num_parts
    pad = s * num_parts - num_nodes
    adj_list = sparse_2d_split(idx, val, (num_nodes, num_nodes), s, 0)
    feats = [features[i:i + s] for i in range(0, num_nodes, s)]
    if pad > 0:
        feats[-1] = np.concatenate([feats[-1], np.zeros((pad, features.shape[1]), features.dtype)])
        labels = np.concatenate([labels, np.zeros((pad,) + labels.shape[1:], labels.dtype)])
        masks = {k: np.concatenate([m, np.zeros(pad, m.dtype)]) for k, m in masks.items()}
        adj_list = [(pi, pv, (s, s * num_parts)) for pi, pv, _ in adj_list]
    return dict(split_size=s, adj=adj_list, features=feats, labels=labels, masks=masks)


def local_view(part, rank):
    """Parted_COO_Graph.__init__ (parted_coo_graph.py:96-111): column split into P blocks [s, s]."""
    pi, pv, shp = part['adj'][rank]
    s = shp[0]
    blocks = sparse_2d_split(pi, pv, shp, s, 1)
    return dict(adj_parts=blocks, features=part['features'][rank],
                local_labels=part['labels'][s * rank:s * (rank + 1)],
                local_train_mask=part['masks']['train_mask'][s * rank:s * (rank + 1)].astype(bool))


# --------------------------------------------------------------------------- the hot loop
def broadcast_aggregate(adj_parts, feature_blocks):
    """models/gcn.py:16-30 seen from one rank: z = sum_src A[rank, src] @ H_src, stages in src order,
    each stage accumulating into z_loc through spmm(A, B, C) with alpha = beta = 1."""
    s, F = feature_blocks[0].shape
    z = np.zeros((adj_parts[0][2][0], F), dtype=np.float32)
    for (pi, pv, shp), h in zip(adj_parts, feature_blocks):
        spmm_coo(pi[0], pi[1], pv, shp[0], shp[1], _f32c(h), z, 1.0, 1.0)
    return z


def layer_forward(adj_parts, feature_blocks, W):
    """DistGCNLayer.forward (gcn.py:35-42): (sum_j A_ij H_j) @ W."""
    z = broadcast_aggregate(adj_parts, feature_blocks)
    return z, z @ W


def layer_backward(adj_parts, grad_out_blocks, features_local, W):
    """DistGCNLayer.backward (gcn.py:45-53), before the all-reduce of grad_weight.
    NB the same (un-transposed) blocks are used, exactly like gcn.py:47."""
    ag = broadcast_aggregate(adj_parts, grad_out_blocks)
    return ag @ W.T, features_local.T @ ag


# --------------------------------------------------------------------------- adaptive broadcast (older trainer)
def _log_softmax(x):
    m = x.max(axis=1, keepdims=True)
    e = x - m
    return (e - np.log(np.exp(e).sum(axis=1, keepdims=True, dtype=np.float32))).astype(np.float32)


def adaptive_reference(adj_parts_by_rank, features_by_rank, labels_by_rank, train_mask_by_rank, weights, epochs,
                       warm_epoch=50, epsilon=0.040, lr=0.01):
    """1D_CAGNET/dist_1d.py:59-158 (broad_func with its caches), :161-203 (GCNFunc), :206-222 (train) and :244-262
    (Adam, lr 0.01, weights from one seed on every rank), all P ranks simulated in lockstep in one process; with three
    weights the model of dist_1d_4layers.py:207-211 (ReLU, ReLU, log_softmax; epsilon 0.10 there, :73).

    Per broadcast type (dist_1d.py:67-142): 'layer1' blocks travel in epoch 0 and are reused afterwards (:68, :96-102);
    'layer2': the OWNER of a block compares it with the copy it last sent (F.l1_loss = mean absolute difference,
    :107-109) and reuses that copy iff cur_epoch >= warm_epoch and diff < epsilon (:109), telling the others with a
    one-float broadcast (:111-114); when a block is reused every rank but the owner multiplies its stored copy, the owner
    its current block (:98-99, :118-119); every other pass ('layer3', the backward passes) always travels (:77-81,
    :139-140). The aggregation multiplies the SAME blocks forward and backward (:190), grad_weight is summed over the
    ranks (:27).

    adj_parts_by_rank[r][i] = (row, col, val, shape) of A[r, i]; returns a list over epochs of dicts with
    'z' (per layer, per rank: the aggregations), 'use2' (per source: was the stored copy used), 'diff2' (per source:
    the owner's mean absolute difference, None before anything was sent), 'loss' (per rank; None where a rank has no
    training node)."""
    P = len(features_by_rank)
    Ws = [_f32c(W).copy() for W in weights]
    L = len(Ws)
    adam = [dict(m=np.zeros_like(W), v=np.zeros_like(W)) for W in Ws]
    b1, b2, eps_adam = 0.9, 0.999, 1e-8
    sent2 = [None] * P                       # cache_layer2[i] on every rank = the block rank i last sent
    out = []

    def aggregate(r, blocks):
        shape = adj_parts_by_rank[r][0][3]
        z = np.zeros((shape[0], blocks[0].shape[1]), dtype=np.float32)
        for i in range(P):
            row, col, val, shape = adj_parts_by_rank[r][i]
            spmm_coo(row, col, val, shape[0], shape[1], _f32c(blocks[i]), z, 1.0, 1.0)
        return z

    for epoch in range(epochs):
        H = [[_f32c(f) for f in features_by_rank]]                      # H[l][r]: input of layer l + 1
        zs, pres, use2, diff2 = [], [], [False] * P, [None] * P
        for l in range(L):
            cur = H[l]
            if l == 1:                                                  # 'layer2': one decision per source, made by its owner
                for i in range(P):
                    if sent2[i] is not None:
                        diff2[i] = float(np.abs(cur[i] - sent2[i]).mean(dtype=np.float32))
                        use2[i] = bool(epoch >= warm_epoch and np.float32(diff2[i]) < np.float32(epsilon))
                    else:
                        use2[i] = bool(epoch >= warm_epoch and epoch % 5 > 0)      # (:75; never true in epoch 0)
                z = [aggregate(r, [cur[i] if (i == r or not use2[i]) else sent2[i] for i in range(P)]) for r in range(P)]
                for i in range(P):
                    if not use2[i]:
                        sent2[i] = cur[i].copy()
            else:                                                       # 'layer1' (constant input: the copies ARE the blocks), 'layer3'
                z = [aggregate(r, cur) for r in range(P)]
            pre = [(z[r] @ Ws[l]).astype(np.float32) for r in range(P)]
            zs.append(z)
            pres.append(pre)
            H.append([np.maximum(p, 0).astype(np.float32) for p in pre] if l + 1 < L else [_log_softmax(p) for p in pre])
        logp = H[L]
        # loss = nll_loss(mean over the rank's training rows) (:215); d loss / d O through log_softmax (:178-181)
        G, losses = [], []
        for r in range(P):
            idx = np.nonzero(train_mask_by_rank[r])[0]
            g = np.zeros_like(logp[r])
            if idx.size:
                y = labels_by_rank[r][idx].astype(np.int64)
                losses.append(float(-logp[r][idx, y].mean(dtype=np.float32)))
                sm = np.exp(logp[r][idx]).astype(np.float32)
                sm[np.arange(idx.size), y] -= 1.0
                g[idx] = sm / np.float32(idx.size)
            else:
                losses.append(None)                                     # fake loss: zero gradient (:218-220)
            G.append(g.astype(np.float32))
        grads = [None] * L
        for l in range(L - 1, -1, -1):                                  # backward passes: always sent, same blocks (:190)
            ag = [aggregate(r, G) for r in range(P)]
            grads[l] = sum((H[l][r].T @ ag[r]).astype(np.float32) for r in range(P))
            if l > 0:
                G = [((ag[r] @ Ws[l].T).astype(np.float32) * (pres[l - 1][r] > 0)).astype(np.float32) for r in range(P)]
        out.append(dict(z=zs, use2=use2, diff2=diff2, loss=losses))
        for W, g, st in zip(Ws, grads, adam):                           # torch.optim.Adam defaults, lr 0.01 (:262)
            t = epoch + 1
            st['m'] = (b1 * st['m'] + (1 - b1) * g).astype(np.float32)
            st['v'] = (b2 * st['v'] + (1 - b2) * g * g).astype(np.float32)
            mhat = st['m'] / np.float32(1 - b1 ** t)
            vhat = st['v'] / np.float32(1 - b2 ** t)
            W -= (np.float32(lr) * mhat / (np.sqrt(vhat) + np.float32(eps_adam))).astype(np.float32)
    return out
