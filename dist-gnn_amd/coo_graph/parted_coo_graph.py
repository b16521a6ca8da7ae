"""Full-graph and per-rank graph containers for the 1D row partition.

API of light_dist_gnn/coo_graph/parted_coo_graph.py (BasicGraph :9-20, GraphCache :23-48,
COO_Graph :51-93, Parted_COO_Graph :96-115): same class names, constructor arguments,
attributes (adj, features, labels, *_mask, num_nodes/edges/classes, local_num_nodes,
local_num_edges, local_labels, local_train_mask, adj_parts) and the same partition arithmetic
(split_size = ceil(N / P), zero padding of the last block, adj blocks declared (s, s*P)).

Differences: cache files hold plain tensors (sparse matrices as indices/values/size) so they load
with ``torch.load(weights_only=True)``; ``COO_Graph`` accepts an in-memory dataset dict; the
partition can be produced in memory (``partition(..., save=False)``) for benchmarks that never
touch the disk.
"""
import os

import torch

from . import datasets, graph_utils


def _pack(d):
    """dict with sparse tensors -> dict of dense tensors / scalars (weights_only-loadable)."""
    out = {}
    for k, v in d.items():
        if isinstance(v, torch.Tensor) and v.is_sparse:
            v = v.coalesce()
            out[k + '__indices'], out[k + '__values'] = v.indices().clone(), v.values().clone()
            out[k + '__size'] = torch.tensor(list(v.shape))
        else:
            out[k] = v.clone() if isinstance(v, torch.Tensor) else v
    return out


def _unpack(d):
    out = {}
    for k, v in d.items():
        if k.endswith('__indices'):
            base = k[:-len('__indices')]
            out[base] = torch.sparse_coo_tensor(v, d[base + '__values'], tuple(d[base + '__size'].tolist()),
                                                is_coalesced=True)
        elif not (k.endswith('__values') or k.endswith('__size')):
            out[k] = v
    return out


class BasicGraph:
    def __init__(self, d, name, device):
        self.name, self.device, self.attr_dict = name, device, d
        self.adj = d['adj'].to(device)
        self.features = d['features'].to(device)
        labels = d['labels'].to(device)
        self.labels = labels.to(torch.float if labels.dim() == 2 else torch.long)
        self.train_mask, self.val_mask, self.test_mask = (d[k].bool().to(device)
                                                          for k in ('train_mask', 'val_mask', 'test_mask'))
        self.num_nodes, self.num_edges, self.num_classes = d['num_nodes'], d['num_edges'], d['num_classes']

    def __repr__(self):
        counts = ','.join(str(int(m.sum())) for m in (self.train_mask, self.val_mask, self.test_mask))
        return f'<COO Graph: {self.name}, |V|: {self.num_nodes}, |E|: {self.num_edges}, masks: {counts}>'


class GraphCache:
    @staticmethod
    def full_graph_path(name, preprocess_for, root=None):
        return os.path.join(root or datasets.data_root, f'{name}_{preprocess_for}_full.coo_graph')

    @staticmethod
    def parted_graph_path(name, preprocess_for, rank, num_parts, root=None):
        dirpath = os.path.join(root or datasets.data_root, f'{name}_{preprocess_for}_{num_parts}_parts')
        os.makedirs(dirpath, exist_ok=True)
        return os.path.join(dirpath, f'part_{rank}_of_{num_parts}.coo_graph')

    @staticmethod
    def save_dict(d, path):
        os.makedirs(os.path.dirname(path), exist_ok=True)
        torch.save(_pack(d), path)

    @staticmethod
    def load_dict(path):
        return _unpack(torch.load(path, weights_only=True))


class BlockList(list):
    """``adj_parts``: the list of column blocks A[rank, j]; ``transposed`` optionally holds the
    blocks (A^T)[rank, j] for a backward pass that is exact on asymmetric graphs; ``slab`` is the
    whole row block A[rank, :] the list was cut from (global column ids), which lets the all-gather
    form of the pipeline aggregate in one SpMM."""
    transposed = None
    slab = None


def csr_arrays(block):
    """int32 CSR (rowptr[n+1], col[nnz]) + values of a coalesced COO block, with torch integer ops so the
    offline partitioner can run on the CPU. Same arrays as dgnn_coo2csr_i32 / dgnn_idx_i64_to_i32 build
    on the device at first use (checked bit-exact in tests/test_host_golden.py)."""
    idx, n = block.indices(), block.size(0)
    if max(block.shape) >= 2 ** 31 or idx.size(1) >= 2 ** 31:
        raise ValueError('block too large for int32 CSR')
    rowptr = torch.zeros(n + 1, dtype=torch.long, device=idx.device)
    rowptr[1:] = torch.cumsum(torch.bincount(idx[0], minlength=n), 0)
    return rowptr.to(torch.int32), idx[1].to(torch.int32), block.values().clone()


def _add_csr(d, key, slab, split_size):
    for j, blk in enumerate(graph_utils.sparse_2d_split(slab, split_size, split_dim=1)):
        rp, col, val = csr_arrays(blk)
        d[f'{key}{j}__rowptr'], d[f'{key}{j}__col'], d[f'{key}{j}__val'] = rp, col, val
        d[f'{key}{j}__shape'] = torch.tensor(list(blk.shape))


def partition_dict(attr_dict, num_parts, padding=True, with_transpose=False, with_csr=False):
    """The arithmetic of COO_Graph.partition (parted_coo_graph.py:63-93) on an in-memory dict:
    returns (split_size, [per-rank attr dict]). ``with_transpose`` also stores the row slab of A^T
    per rank ('adj_t') -- the role of the older partitioner's pre-transposed column slabs
    (1D_CAGNET/dist_data.py:59-63,84-108). ``with_csr`` stores, next to the COO slab,
    the int32 CSR of every column block A[rank, j] ('csr{j}__rowptr/col/val', and 'csrt{j}__*' for the
    transposed blocks), so loading a part needs no index conversion at all -- the reference converts
    int64 -> int32 twice per SpMM call (gcn.py:11)."""
    n = attr_dict['num_nodes']
    adj, feats = attr_dict['adj'], attr_dict['features']
    split_size = (n + num_parts - 1) // num_parts
    pad = split_size * num_parts - n
    shared = {k: v for k, v in attr_dict.items() if k not in ('adj', 'features')}
    adj_rows = graph_utils.sparse_2d_split(adj, split_size)
    adj_t_rows = graph_utils.sparse_2d_split(adj.t().coalesce(), split_size) if with_transpose else None
    feat_rows = list(torch.split(feats, split_size))
    while len(adj_rows) < num_parts:       # more ranks than non-empty row slabs (tiny graphs)
        adj_rows.append(torch.sparse_coo_tensor(torch.zeros(2, 0, dtype=torch.long, device=adj.device),
                                                torch.zeros(0, dtype=adj.dtype, device=adj.device),
                                                (split_size, adj.size(1)), is_coalesced=True))
        feat_rows.append(feats.new_zeros(0, feats.size(1)))
    if padding and pad > 0:
        short = split_size - feat_rows[-1].size(0)
        feat_rows[-1] = torch.cat([feat_rows[-1], feats.new_zeros(short, feats.size(1))])
        for i in range(len(feat_rows) - 1):
            if feat_rows[i].size(0) < split_size:
                feat_rows[i] = torch.cat([feat_rows[i], feats.new_zeros(split_size - feat_rows[i].size(0), feats.size(1))])
        lab = shared['labels']
        shared['labels'] = torch.cat([lab, lab.new_zeros((pad,) + tuple(lab.shape[1:]))])
        for key in ('train_mask', 'val_mask', 'test_mask'):
            shared[key] = torch.cat([shared[key], shared[key].new_zeros(pad)])
        adj_rows = [torch.sparse_coo_tensor(a.indices(), a.values(), (split_size, split_size * num_parts),
                                            is_coalesced=True) for a in adj_rows]
        if adj_t_rows is not None:
            adj_t_rows = [torch.sparse_coo_tensor(a.indices(), a.values(), (split_size, split_size * num_parts),
                                                  is_coalesced=True) for a in adj_t_rows]
    parts = []
    for i in range(num_parts):
        d = dict(shared)
        d['adj'], d['features'] = adj_rows[i], feat_rows[i]
        if adj_t_rows is not None and i < len(adj_t_rows):
            d['adj_t'] = adj_t_rows[i]
        if with_csr:
            _add_csr(d, 'csr', adj_rows[i], split_size)
            if 'adj_t' in d:
                _add_csr(d, 'csrt', d['adj_t'], split_size)
        parts.append(d)
    return split_size, parts


class COO_Graph(BasicGraph):
    def __init__(self, name, preprocess_for='GCN', full_graph_cache_enabled=True, device='cpu', attr_dict=None):
        self.preprocess_for = preprocess_for
        self.cache_path = GraphCache.full_graph_path(name, preprocess_for)
        if attr_dict is not None:                       # already-loaded raw dataset dict
            cached = graph_utils.preprocess(name, dict(attr_dict), preprocess_for)
        elif full_graph_cache_enabled and os.path.exists(self.cache_path):
            cached = GraphCache.load_dict(self.cache_path)
        else:
            raw = datasets.load_dataset(name)
            raw['edge_index'] = raw['edge_index'].long()
            cached = graph_utils.preprocess(name, raw, preprocess_for)
            if full_graph_cache_enabled:
                GraphCache.save_dict(cached, self.cache_path)
        super().__init__(cached, name, device)

    def partition(self, num_parts, padding=True, save=True, with_transpose=False, with_csr=False):
        d = dict(self.attr_dict)
        d.update(adj=self.adj, features=self.features)
        _, parts = partition_dict(d, num_parts, padding, with_transpose, with_csr)
        if save:
            for i, part in enumerate(parts):
                GraphCache.save_dict(part, GraphCache.parted_graph_path(self.name, self.preprocess_for, i, num_parts))
        return parts


class Parted_COO_Graph(BasicGraph):
    def __init__(self, name, rank, num_parts, preprocess_for='GCN', device='cpu', attr_dict=None,
                 transposed_backward=False):
        self.rank, self.num_parts = rank, num_parts
        if attr_dict is None:
            path = GraphCache.parted_graph_path(name, preprocess_for, rank, num_parts)
            if not os.path.exists(path):
                raise Exception('Not parted yet. Run COO_Graph.partition() first.', path)
            attr_dict = GraphCache.load_dict(path)
        super().__init__(attr_dict, name, device)
        self.local_num_nodes = self.adj.size(0)
        self.local_num_edges = self.adj.values().size(0)
        lo, hi = self.local_num_nodes * rank, self.local_num_nodes * (rank + 1)
        self.local_labels = self.labels[lo:hi]
        self.local_train_mask = self.train_mask[lo:hi].bool()
        self.adj_parts = BlockList(graph_utils.sparse_2d_split(self.adj, self.local_num_nodes, split_dim=1))
        self.adj_parts.slab = self.adj
        if transposed_backward:
            if 'adj_t' not in attr_dict:
                raise Exception('transposed_backward needs part files written with partition(..., with_transpose=True)')
            adj_t = attr_dict['adj_t'].to(device)
            self.adj_parts.transposed = BlockList(graph_utils.sparse_2d_split(adj_t, self.local_num_nodes, split_dim=1))
            self.adj_parts.transposed.slab = adj_t
        self.csr_from_file = self._attach_csr(attr_dict, 'csr', self.adj_parts)
        if self.adj_parts.transposed is not None:
            self._attach_csr(attr_dict, 'csrt', self.adj_parts.transposed)

    def _attach_csr(self, attr_dict, key, blocks):
        """Part files written with partition(..., with_csr=True): hand the stored int32 CSR of every block
        to the SpMM operator (spmm_cpp.CsrBlock) so nothing is converted at the first call. GPU only."""
        if f'{key}0__rowptr' not in attr_dict or torch.device(self.device).type != 'cuda':
            return False
        import spmm_cpp
        for j, blk in enumerate(blocks):
            shape = tuple(int(x) for x in attr_dict[f'{key}{j}__shape'])
            if shape != tuple(blk.shape):
                raise Exception(f'part file: stored CSR block {j} has shape {shape}, the COO block {tuple(blk.shape)}')
            blk._dgnn_csr = spmm_cpp.CsrBlock(shape[0], shape[1], *(attr_dict[f'{key}{j}__{a}'].to(self.device).contiguous()
                                                                   for a in ('rowptr', 'col', 'val')))
        return True

    def __repr__(self):
        return super().__repr__() + f'<Local: {self.rank}, |V|: {self.local_num_nodes}, |E|: {self.local_num_edges}>'
