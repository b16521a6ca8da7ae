"""spmm_cpp -- drop-in for the reference's torch extension of the same import name.

Reference boundary (paths relative to the reference repository):
  light_dist_gnn/spmm_cpp/spmm.cpp:16-17,35-37   spmm_cusparse(row, col, val, n, m, B, C, alpha, beta)
  1D_CAGNET/sparse-extension/sparse_coo_tensor.cpp:115-121  spmm_gpu(row, col, val, n, m, B, C)
  light_dist_gnn/models/gcn.py:9-13              `from spmm_cpp import spmm_cusparse`

This module is a thin ctypes stub over the C ABI of ``lib/libdgnn_spmm.so``
(include/dgnn_spmm.h): it passes raw device pointers, sizes and torch's current
HIP stream. PyTorch is plumbing here (device memory + streams); all arithmetic
is in the hand-written gfx950 kernels.

Differences from the reference, on purpose:
  * errors raise ``RuntimeError`` instead of being printf'd and ignored (spmm.cpp:7-14);
  * there is NO CPU fallback: a missing library is an error at import (not an
    ImportError, so ``try: from spmm_cpp import ...; except ImportError`` in the
    reference's models/gcn.py:9-13 cannot silently fall back to ``addmm_``), and
    tensors that are not on a GPU are rejected;
  * ``CsrBlock`` lets a caller convert a block to int32 CSR once instead of
    re-marshalling int64 indices on every call (gcn.py:11).
"""
import ctypes
import math
import os

import torch

_LIB_PATH = os.environ.get('DGNN_LIB') or os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'lib', 'libdgnn_spmm.so')   # DGNN_LIB: A/B runs of two builds
if not os.path.exists(_LIB_PATH):
    raise RuntimeError(
        f'spmm_cpp: {os.path.normpath(_LIB_PATH)} is missing -- build it with '
        '`python -c "import __graft_entry__ as g; g.build()"` (hipcc --offload-arch=gfx950). '
        'There is no CPU fallback.')
_lib = ctypes.CDLL(_LIB_PATH)

_p, _i32, _i64, _f32 = ctypes.c_void_p, ctypes.c_int32, ctypes.c_int64, ctypes.c_float


class _CPlan(ctypes.Structure):          # struct dgnn_plan (include/dgnn_spmm.h)
    _fields_ = [('nstage', _i32), ('seg', _p), ('stage_seg_off', _p), ('cut', _p),
                ('stage_cut_off', _p), ('npartial', _i64), ('max_seg_len', _i32), ('covers_rows', _i32),
                ('hot_rows', _i32), ('hot_cols', _p), ('seg_hot', _p)]


class _CBundlePlan(ctypes.Structure):    # struct dgnn_bundle_plan (include/dgnn_spmm.h)
    _fields_ = [('nstage', _i32), ('rows_per_bundle', _i32), ('desc', _p), ('stage_desc_off', _p), ('ucol', _p), ('uval', _p),
                ('cut', _p), ('stage_cut_off', _p), ('npartial', _i64)]


class _CDensePlan(ctypes.Structure):     # struct dgnn_dense_plan (include/dgnn_spmm.h)
    _fields_ = [('rows_per_tile', _i32), ('ntile', _i32), ('npiece', _i32), ('ncut', _i32), ('desc', _p), ('rows', _p),
                ('kcol', _p), ('kval', _p), ('cut', _p), ('nk', _i64), ('npartial', _i64)]


class _CStripPlan(ctypes.Structure):     # struct dgnn_strip_plan (include/dgnn_spmm.h)
    _fields_ = [('rows_per_tile', _i32), ('strip_cols', _i32), ('waves', _i32), ('line_entries', _i32), ('ntile', _i32),
                ('npiece', _i32), ('ncut', _i32), ('reserved', _i32),
                ('desc', _p), ('rows', _p), ('ucol', _p), ('ent', _p), ('ebase', _p), ('cut', _p),
                ('nu', _i64), ('nline', _i64), ('npartial', _i64)]


class _CLayout(ctypes.Structure):        # struct dgnn_plan_layout (include/dgnn_spmm.h)
    _fields_ = [('nstage', _i32), ('piece', _i32), ('hdr_ints', _i64), ('nseg_max', _i64), ('ncut_max', _i64),
                ('npartial_max', _i64), ('ws_ints', _i64)]


_pi64 = ctypes.POINTER(ctypes.c_int64)
_SIGNATURES = {                          # argument types, in the order of include/dgnn_spmm.h; all return int
    'dgnn_version': [],
    'dgnn_prof_begin': [],
    'dgnn_prof_end': [ctypes.POINTER(ctypes.c_double), _pi64, ctypes.POINTER(ctypes.c_double), _pi64],
    'dgnn_prof_end_by_kernel': [ctypes.POINTER(ctypes.c_double), _pi64, _i32],
    'dgnn_coo2csr_i32': [_p, _i64, _i32, _p, _p],
    'dgnn_idx_i64_to_i32': [_p, _i64, _p, _p],
    'dgnn_pad_rows_f32': [_p, _i64, _p, _i64, _i32, _i32, _p],
    'dgnn_gather_rows_f32': [_p, _i64, _i32, _p, _i32, _p, _i64, _i32, _p],
    'dgnn_spmm_csr_f32': [_p, _p, _p, _i32, _i32, _i64, _p, _i64, _p, _i64, _i32, _f32, _f32, _p],
    'dgnn_spmm_coo_f32': [_p, _p, _p, _i32, _i32, _i64, _p, _p, _i64, _p, _i64, _i32, _f32, _f32, _p],
    'dgnn_coo_aligned_workspace': [_i32, _i32, _i64, _i64, _pi64, _pi64],
    'dgnn_spmm_coo_aligned_f32': [_p, _p, _p, _i32, _i32, _i64, _p, _p, _p, _i64, _p, _i64, _p, _i64, _i32, _f32, _f32, _p],
    'dgnn_noalloc_limits': [_pi64],
    'dgnn_noalloc_scratch_needed': [_i32, _i32, _i64, _i64, _pi64],
    'dgnn_spmm_coo_noalloc_f32': [_p, _p, _p, _i32, _i32, _i64, _p, _i64, _p, _i64, _i32, _f32, _f32, _p],
    'dgnn_plan_layout_for': [_i32, _i32, _i64, _i64, _i32, _i32, ctypes.POINTER(_CLayout)],
    'dgnn_plan_build': [_p, _p, _i32, _i32, ctypes.POINTER(_CLayout), _p, _p],
    'dgnn_spmm_plan_f32': [ctypes.POINTER(_CPlan), _p, _p, _i32, _i32, _p, _i64, _p, _i64, _p, _i64, _i32, _f32, _f32, _p],
    'dgnn_spmm_bundle_f32': [ctypes.POINTER(_CBundlePlan), _i32, _i32, _p, _i64, _p, _i64, _p, _i64, _i32, _f32, _f32, _p],
    'dgnn_spmm_dense_f32': [ctypes.POINTER(_CDensePlan), _i32, _i32, _p, _i64, _p, _i64, _p, _i64, _i32, _f32, _p],
    'dgnn_spmm_strip_f32': [ctypes.POINTER(_CStripPlan), _i32, _i32, _p, _i64, _p, _i64, _p, _i64, _i32, _f32, _p],
    'dgnn_spmm_strip_overwrite_f32': [ctypes.POINTER(_CStripPlan), _i32, _i32, _p, _i64, _p, _i64, _p, _i64, _i32, _f32, _p],
    'dgnn_spmm_plan_overwrite_f32': [ctypes.POINTER(_CPlan), _p, _p, _i32, _i32, _p, _i64, _p, _i64, _p, _i64, _i32, _f32, _p],
}
for _name, _args in _SIGNATURES.items():
    getattr(_lib, _name).argtypes = _args
    getattr(_lib, _name).restype = ctypes.c_int
_lib.dgnn_last_error.restype = ctypes.c_char_p
_ABI = 280                                # struct layouts and entry points of include/dgnn_spmm.h this stub was written for
if _lib.dgnn_version() != _ABI:
    raise RuntimeError(f'spmm_cpp: {os.path.normpath(_LIB_PATH)} is version {_lib.dgnn_version()}, this stub needs {_ABI}: '
                       'rebuild it (python -c "import __graft_entry__ as g; g.build(force=True)")')

lib = _lib  # exposed for tests that call the C ABI directly


def version():
    return _lib.dgnn_version()


def _check(rc, what):
    if rc != 0:
        raise RuntimeError(f'{what}: {_lib.dgnn_last_error().decode()} (status {rc})')


def _stream(t):
    return ctypes.c_void_p(torch.cuda.current_stream(t.device).cuda_stream)


def _need(t, dtype, name, ndim):
    if not isinstance(t, torch.Tensor):
        raise RuntimeError(f'spmm_cpp: {name} must be a torch.Tensor')
    if not t.is_cuda:
        raise RuntimeError(f'spmm_cpp: {name} is on {t.device}; this operator has no CPU path')
    if t.dtype != dtype:
        raise RuntimeError(f'spmm_cpp: {name} must be {dtype}, got {t.dtype}')
    if t.dim() != ndim:
        raise RuntimeError(f'spmm_cpp: {name} must have {ndim} dims, got {t.dim()}')


def _dense(t, name):
    """Row-major 2-D fp32 view description: (ptr, ld). Rows must be unit-stride."""
    _need(t, torch.float32, name, 2)
    if t.size(1) > 1 and t.stride(1) != 1:
        raise RuntimeError(f'spmm_cpp: {name} must be row-major (stride(1) == 1)')
    ld = t.stride(0) if t.size(0) > 1 else max(t.size(1), 1)
    if ld < t.size(1):
        raise RuntimeError(f'spmm_cpp: {name} has overlapping rows')
    return t.data_ptr(), ld


_FAST_COO_MIN_NNZ = 1 << 16
_scratch_cache = {}


def _scratch(kind, count, dtype, dev):
    """Reusable workspace per (kind, device, stream), grown on demand: calls on one stream are ordered."""
    key = (kind, dev, torch.cuda.current_stream(dev).cuda_stream)
    buf = _scratch_cache.get(key)
    if buf is None or buf.numel() < count:
        buf = _scratch_cache[key] = torch.empty(max(int(count), 1), dtype=dtype, device=dev)
    return buf


def spmm_cusparse(A_row_idx, A_col_idx, A_values, A_row, A_col, B, C, alpha, beta):
    """C <- alpha * A @ B + beta * C in place; A is row-sorted COO (int32 indices, fp32 values).

    Same positional signature as the reference's pybind export (spmm.cpp:17).
    """
    _need(A_row_idx, torch.int32, 'A_row_idx', 1)
    _need(A_col_idx, torch.int32, 'A_col_idx', 1)
    _need(A_values, torch.float32, 'A_values', 1)
    nnz = A_values.numel()
    if A_row_idx.numel() != nnz or A_col_idx.numel() != nnz:
        raise RuntimeError('spmm_cpp: row/col/values lengths differ')
    for t, nm in ((A_row_idx, 'A_row_idx'), (A_col_idx, 'A_col_idx'), (A_values, 'A_values')):
        if not t.is_contiguous():
            raise RuntimeError(f'spmm_cpp: {nm} must be contiguous')
    bptr, ldb = _dense(B, 'B')
    cptr, ldc = _dense(C, 'C')
    n, m, F = int(A_row), int(A_col), C.size(1)
    if not (0 <= n < 2 ** 31 and 0 <= m < 2 ** 31 and nnz < 2 ** 31):
        raise RuntimeError('spmm_cpp: block exceeds int32 indexing')
    if B.size(0) != m or C.size(0) != n or B.size(1) != F:
        raise RuntimeError(f'spmm_cpp: shape mismatch A[{n}x{m}] B{tuple(B.shape)} C{tuple(C.shape)}')
    if not (A_values.device == A_row_idx.device == A_col_idx.device == B.device == C.device):
        raise RuntimeError('spmm_cpp: operands are on different devices')
    with torch.cuda.device(C.device):
        ws = torch.empty(n + 1, dtype=torch.int32, device=C.device)
        if nnz >= _FAST_COO_MIN_NNZ and F >= 8:
            # big enough to pay for a padded copy of B: the staged kernel on a one-run-per-row work list
            Bp = B if (ldb % 4 == 0 and B.data_ptr() % 16 == 0) else pad_rows(B, _pad_buffer(m, padded_width(F), B.device))
            ws_ints, prow = ctypes.c_int64(), ctypes.c_int64()
            ldp = Bp.stride(0)
            _check(_lib.dgnn_coo_aligned_workspace(n, m, nnz, ldp, ctypes.byref(ws_ints), ctypes.byref(prow)), 'coo_aligned_workspace')
            plan_ws = _scratch('plan', ws_ints.value, torch.int32, C.device)
            partial = _scratch('partial', prow.value * ldp, torch.float32, C.device)
            rc = _lib.dgnn_spmm_coo_aligned_f32(A_row_idx.data_ptr(), A_col_idx.data_ptr(), A_values.data_ptr(),
                                                n, m, nnz, ws.data_ptr(), plan_ws.data_ptr(), Bp.data_ptr(), ldp,
                                                cptr, ldc, partial.data_ptr(), ldp, F, float(alpha), float(beta), _stream(C))
        else:
            rc = _lib.dgnn_spmm_coo_f32(A_row_idx.data_ptr(), A_col_idx.data_ptr(), A_values.data_ptr(),
                                        n, m, nnz, ws.data_ptr(), bptr, ldb, cptr, ldc, F,
                                        float(alpha), float(beta), _stream(C))
    _check(rc, 'spmm_cusparse')


def spmm_gpu(A_row_idx, A_col_idx, A_values, A_row, A_col, B, C):
    """Older ABI (sparse_coo_tensor.cpp:115-196): alpha = beta = 1."""
    spmm_cusparse(A_row_idx, A_col_idx, A_values, A_row, A_col, B, C, 1.0, 1.0)


class CsrBlock:
    """One adjacency block A[n x m] held on the GPU as int32 CSR + fp32 values.

    Built once per block (per process) from a coalesced ``torch.sparse_coo`` tensor,
    replacing the per-call ``A.indices()[k].int()`` copies of gcn.py:11 and the
    per-call coo2csr of spmm_original.cpp:141-146.
    """

    def __init__(self, n, m, rowptr, col, val):
        self.n, self.m = int(n), int(m)
        self.rowptr, self.col, self.val = rowptr, col, val
        self.nnz = val.numel()
        self.device = val.device
        self._distinct = None

    def require_distinct(self, who):
        """The tile / bundle / strip builders scatter one value per (row, column) into their lists: an entry that
        occurs twice would silently lose one of its values. Blocks from a coalesced tensor (from_coo) are known to be
        fine; a hand-built block is checked once."""
        if self._distinct is None:
            if self.nnz > 1:
                rp = self.rowptr.long()
                rows = torch.repeat_interleave(torch.arange(self.n, device=self.device), rp[1:] - rp[:-1])
                self._distinct = int(torch.unique(rows * self.m + self.col.long()).numel()) == self.nnz
            else:
                self._distinct = True
        if not self._distinct:
            raise RuntimeError(f'{who}: the block lists some (row, column) more than once; coalesce it first '
                               '(CsrBlock.from_coo takes a coalesced tensor)')

    @classmethod
    def from_coo(cls, A):
        if not A.is_sparse:
            raise RuntimeError('CsrBlock.from_coo: expected a torch.sparse_coo tensor')
        if not A.is_coalesced():
            raise RuntimeError('CsrBlock.from_coo: tensor must be coalesced (row-sorted)')
        idx, val = A.indices(), A.values()
        _need(val, torch.float32, 'A.values()', 1)
        n, m = A.size(0), A.size(1)
        nnz = val.numel()
        if max(n, m) >= 2 ** 31 or nnz >= 2 ** 31:
            raise RuntimeError('CsrBlock.from_coo: block exceeds int32 indexing')
        dev = val.device
        with torch.cuda.device(dev):
            row32 = torch.empty(nnz, dtype=torch.int32, device=dev)
            col32 = torch.empty(nnz, dtype=torch.int32, device=dev)
            rowptr = torch.empty(n + 1, dtype=torch.int32, device=dev)
            idx = idx.contiguous()
            s = _stream(val)
            _check(_lib.dgnn_idx_i64_to_i32(idx[0].data_ptr(), nnz, row32.data_ptr(), s), 'idx_i64_to_i32')
            _check(_lib.dgnn_idx_i64_to_i32(idx[1].data_ptr(), nnz, col32.data_ptr(), s), 'idx_i64_to_i32')
            _check(_lib.dgnn_coo2csr_i32(row32.data_ptr(), nnz, n, rowptr.data_ptr(), s), 'coo2csr')
        blk = cls(n, m, rowptr, col32, val.contiguous())
        blk._distinct = True                                # coalesced: every (row, column) once
        return blk


def spmm_csr(A, B, C, alpha=1.0, beta=1.0):
    """C <- alpha * A @ B + beta * C in place for a ``CsrBlock`` A."""
    bptr, ldb = _dense(B, 'B')
    cptr, ldc = _dense(C, 'C')
    F = C.size(1)
    if B.size(0) != A.m or C.size(0) != A.n or B.size(1) != F:
        raise RuntimeError(f'spmm_csr: shape mismatch A[{A.n}x{A.m}] B{tuple(B.shape)} C{tuple(C.shape)}')
    if not (A.device == B.device == C.device):
        raise RuntimeError('spmm_csr: operands are on different devices')
    with torch.cuda.device(C.device):
        rc = _lib.dgnn_spmm_csr_f32(A.rowptr.data_ptr(), A.col.data_ptr(), A.val.data_ptr(),
                                    A.n, A.m, A.nnz, bptr, ldb, cptr, ldc, F,
                                    float(alpha), float(beta), _stream(C))
    _check(rc, 'spmm_csr')


def prof_begin():
    _check(_lib.dgnn_prof_begin(), 'prof_begin')


def prof_end():
    """-> dict(stage_ms, stage_launches, other_ms, other_launches) since prof_begin()."""
    a, b = ctypes.c_double(), ctypes.c_double()
    na, nb = ctypes.c_int64(), ctypes.c_int64()
    _check(_lib.dgnn_prof_end(ctypes.byref(a), ctypes.byref(na), ctypes.byref(b), ctypes.byref(nb)), 'prof_end')
    return dict(stage_ms=a.value, stage_launches=na.value, other_ms=b.value, other_launches=nb.value)


PROF_KERNELS = ('stage', 'bundle', 'dense', 'strip', 'hot', 'rows', 'merge', 'helper')     # DGNN_PROF_* of the header


def prof_end_by_kernel():
    """-> {kernel: (ms, launches)} since prof_begin(), kernels as in PROF_KERNELS."""
    n = len(PROF_KERNELS)
    ms, cnt = (ctypes.c_double * n)(), (ctypes.c_int64 * n)()
    _check(_lib.dgnn_prof_end_by_kernel(ms, cnt, n), 'prof_end_by_kernel')
    return {k: (ms[i], cnt[i]) for i, k in enumerate(PROF_KERNELS)}


def set_variant(v):
    """A/B knob of the benchmark tools: only in a library built with -DDGNN_EXPERIMENTS (include/dgnn_spmm_experiments.h;
    DGNN_LIB=.../lib/libdgnn_spmm_exp.so). The product library has no such switch."""
    if not hasattr(_lib, 'dgnn_set_variant'):
        if int(v) == 0:
            return
        raise RuntimeError('spmm_cpp.set_variant: this is the product library; load the experiments build '
                           '(DGNN_LIB=<repo>/dist-gnn_amd/lib/libdgnn_spmm_exp.so) for the A/B knobs')
    _lib.dgnn_set_variant.argtypes = [ctypes.c_int]
    _check(_lib.dgnn_set_variant(int(v)), 'set_variant')


# --------------------------------------------------------------------------- fast path: work plan
PIECE_LEN = 4096             # longest run one lane group walks (strip by strip); longer runs are cut and merged


def padded_width(F):
    """Leading dimension that puts every feature row on a 128-byte line."""
    return (F + 31) // 32 * 32


def pad_rows(src, dst=None):
    """Copy a [n, F] fp32 matrix into a zero-padded [n, padded_width(F)] buffer (returns the buffer)."""
    sptr, lds = _dense(src, 'src')
    n, F = src.shape
    if dst is None:
        dst = torch.empty(n, padded_width(F), dtype=torch.float32, device=src.device)
    dptr, ldd = _dense(dst, 'dst')
    if dst.size(0) != n or dst.size(1) < F:
        raise RuntimeError('pad_rows: dst has the wrong shape')
    with torch.cuda.device(src.device):
        _check(_lib.dgnn_pad_rows_f32(sptr, lds, dptr, ldd, n, F, _stream(src)), 'pad_rows')
    return dst


def is_row_padded(t):
    """True when the kernels can read ``t`` ([n, F] fp32 on the GPU) in place as a B operand: unit column stride,
    rows at the padded pitch, first row on a 128-byte line. (A contiguous matrix whose width is a multiple of 32
    floats qualifies as it is.)"""
    return (t.dim() == 2 and t.is_cuda and t.dtype == torch.float32 and t.stride(1) == 1
            and t.stride(0) == padded_width(t.size(1)) and t.data_ptr() % 128 == 0)


def row_padded(t):
    """``t`` stored at the kernels' row pitch: a [n, F] view of a zero-filled [n, padded_width(F)] buffer holding
    t's values (t itself when it already qualifies). For matrices that are aggregated again and again unchanged --
    the dataset's input features -- so that no epoch has to copy them into a padded block first."""
    if is_row_padded(t):
        return t
    if not (t.dim() == 2 and t.is_cuda and t.dtype == torch.float32):
        raise RuntimeError('row_padded: expected a 2-D fp32 CUDA tensor')
    n, F = t.shape
    buf = torch.zeros(n, padded_width(F), dtype=torch.float32, device=t.device)
    view = buf[:, :F]
    view.copy_(t)
    return view


def gather_rows(src, idx, dst=None):
    """dst[r] = src[idx[r]] packed at the padded pitch (zero fill): the rows of a feature block one peer
    needs (1D_CAGNET/dist_1d.py:43 ``t[needed_rows_idx]``). ``idx`` int32 on the device, values in
    [0, src.size(0)) -- the caller guarantees the range (it built the list from the adjacency block)."""
    sptr, lds = _dense(src, 'src')
    n, F = src.shape
    _need(idx, torch.int32, 'idx', 1)
    if idx.device != src.device:
        raise RuntimeError('gather_rows: idx and src are on different devices')
    idx = idx.contiguous()
    k = idx.numel()
    if dst is None:
        dst = torch.empty(k, padded_width(F), dtype=torch.float32, device=src.device)
    dptr, ldd = _dense(dst, 'dst')
    if dst.size(0) < k or dst.size(1) < F:
        raise RuntimeError('gather_rows: dst is too small')
    with torch.cuda.device(src.device):
        _check(_lib.dgnn_gather_rows_f32(sptr, lds, n, idx.data_ptr(), k, dptr, ldd, F, _stream(src)), 'gather_rows')
    return dst


def default_stage_count(n, m, nnz, F):
    """Column-block stages for a block (measured on MI355X, sweeps on Reddit- and products-shaped blocks of
    the 1/2/4/8-rank partitions): enough stages that one 256-byte feature chunk of one column
    block is ~16 MiB of B (~32 MiB for blocks at most 64 floats wide, which run at the L2 gather ceiling out
    of the Infinity Cache anyway), never so many that a (row, column block) run averages under ~24
    nonzeros, at most 16."""
    ld = padded_width(F)
    if n == 0 or nnz == 0:
        return 1
    budget = (16 << 20) if ld > 64 else (32 << 20)
    by_cache = -(-m * 256 // budget)
    by_density = max(1, int(nnz / max(n, 1) / 24))
    return int(max(1, min(by_cache, by_density, 16)))


def default_piece_len(nnz, nstage):
    """Longest run handed to one lane group. Longer pieces mean fewer partial rows to merge and more rows
    with the oracle's exact chain; shorter ones balance small blocks. Measured optimum ~ nnz per stage /
    7000 (256-512 for a 1.8 M-nonzero block, 4096 from ~30 M up), as a power of two in [128, PIECE_LEN]."""
    want = max(1.0, nnz / max(nstage, 1) / 7000.0)
    p = 128
    while p * 1.41 < want and p < PIECE_LEN:
        p *= 2
    return p


class SegPlan:
    """Host-side builder of ``struct dgnn_plan`` for one CsrBlock (integer work, done once).

    Layout and meaning of every array: include/dgnn_spmm.h. Built with torch index ops on the
    block's device; the result is deterministic for a given block, stage count and piece length.
    """

    def __init__(self, A, nstage=1, piece_len=None, hot_rows=0, row_bucket=None, skip_rows=None):
        """``skip_rows`` (bool [n]): rows another call writes (the rows of row-strip tiles, all empty in this block):
        they are not listed as empty runs, so a product that overwrites C leaves them alone.
        ``hot_rows`` = 256 or 512: a hot-tile plan (struct dgnn_plan, hot_* fields) -- per stage the hot_rows
        most referenced columns of the column block are served from LDS; the plan then carries its own entry
        arrays ``col`` / ``val`` (hot entries of every run first, column replaced by the tile slot)."""
        dev = A.device
        rp = A.rowptr.long()
        n, nnz = A.n, A.nnz
        self.n = n
        self.row_bucket = int(row_bucket or 0)
        self.hot_rows = int(hot_rows) if nnz else 0
        if self.hot_rows not in (0, 256, 512):
            raise ValueError('hot_rows must be 0, 256 or 512')
        self.col, self.val = A.col, A.val                     # the entry arrays this plan indexes
        if piece_len is None:
            piece_len = int(os.environ.get('DGNN_PIECE_LEN', 0)) or default_piece_len(A.nnz, nstage)
        if not 1 <= piece_len < 2 ** 31:
            raise ValueError('piece_len must be positive')
        nstage = max(1, int(nstage))
        cbw = max(1, -(-A.m // nstage))
        self.nstage, self.piece_len = nstage, piece_len
        i64 = dict(dtype=torch.long, device=dev)
        covers = 1 if nnz else 0
        if nnz == 0:
            self.seg = torch.zeros(0, 4, dtype=torch.int32, device=dev)
            if skip_rows is not None:                             # nothing to multiply, but the rows that are ours get their zeros
                mine = torch.nonzero(~skip_rows).flatten()
                z = torch.zeros_like(mine)
                self.seg = torch.stack([mine, z, z, z - 1], dim=1).int().contiguous()
                covers = 1
            self.cut = torch.zeros(0, 4, dtype=torch.int32, device=dev)
            self.stage_seg_off = torch.zeros(nstage + 1, dtype=torch.long)
            self.stage_seg_off[1:] = self.seg.size(0)
            self.stage_cut_off = torch.zeros(nstage + 1, dtype=torch.long)
            self.npartial = 0
        else:
            row_idx = torch.repeat_interleave(torch.arange(n, **i64), rp[1:] - rp[:-1])
            cb = A.col.long() // cbw
            key = row_idx * nstage + cb                       # non-decreasing in CSR order
            first = torch.ones(nnz, dtype=torch.bool, device=dev)
            first[1:] = key[1:] != key[:-1]
            k0 = torch.nonzero(first).flatten()
            k1 = torch.cat([k0[1:], torch.tensor([nnz], **i64)])
            run_row, run_cb = row_idx[k0], cb[k0]
            run_hot = None
            if self.hot_rows:
                T = self.hot_rows
                cnt = torch.bincount(A.col.long(), minlength=A.m)
                cstage = torch.arange(A.m, **i64) // cbw
                order = torch.argsort(cstage * (nnz + 1) + (nnz - cnt), stable=True)      # per stage, most referenced first
                pos = torch.arange(A.m, **i64) - (cstage[order] * cbw)                     # rank inside the stage
                take = (pos < T) & (cnt[order] > 0)
                slot = torch.full((A.m,), -1, **i64)
                slot[order[take]] = pos[take]
                hot_cols = torch.zeros(nstage * T, **i64)
                hot_cols[cstage[order[take]] * T + pos[take]] = order[take]
                self.hot_cols = hot_cols.int().contiguous()
                s_of = slot[A.col.long()]
                is_hot = s_of >= 0
                perm = torch.argsort(key * 2 + (~is_hot).long(), stable=True)              # inside a run: hot first, order kept
                self.col = torch.where(is_hot, s_of, A.col.long())[perm].int().contiguous()
                self.val = A.val[perm].contiguous()
                self.perm = perm                                  # tests: entry order of the plan in terms of A's
                hot_before = torch.cumsum(is_hot.long(), 0) - is_hot.long()                # exclusive, in CSR order
                total_hot = torch.cat([hot_before, hot_before[-1:] + is_hot[-1:].long()])
                run_hot = total_hot[k1] - total_hot[k0]           # hot entries of every run
                self.hot_fraction = float(is_hot.sum().item()) / nnz
                del cnt, cstage, order, pos, take, slot, s_of, is_hot, hot_before, total_hot
            del row_idx, cb, key, first
            pieces = (k1 - k0 + piece_len - 1) // piece_len
            is_cut = pieces > 1
            run_of = torch.repeat_interleave(torch.arange(k0.numel(), **i64), pieces)
            first_piece = torch.cumsum(pieces, 0) - pieces
            pidx = torch.arange(run_of.numel(), **i64) - first_piece[run_of]
            pk0 = k0[run_of] + pidx * piece_len
            pk1 = torch.minimum(pk0 + piece_len, k1[run_of])
            prow, pcb, pcut = run_row[run_of], run_cb[run_of], is_cut[run_of]
            pout = torch.where(pcut, torch.cumsum(pcut.long(), 0) - 1, torch.full_like(pk0, -1))
            pkh = None
            if run_hot is not None:                               # where a piece's hot entries end
                pkh = torch.minimum(torch.maximum((k0 + run_hot)[run_of], pk0), pk1)
            # rows with nothing in the first column block are listed there as empty runs: the first stage
            # then stores every row and a product that overwrites C needs no zero fill (covers_rows)
            has0 = torch.zeros(n, dtype=torch.bool, device=dev)
            has0[run_row[run_cb == 0]] = True
            if skip_rows is not None:
                if bool((skip_rows & (rp[1:] > rp[:-1])).any()):
                    raise ValueError('SegPlan: skip_rows must be empty rows of this block')
                has0 |= skip_rows
            miss = torch.nonzero(~has0).flatten()
            if miss.numel():
                at = rp[miss]
                prow, pk0, pk1 = torch.cat([prow, miss]), torch.cat([pk0, at]), torch.cat([pk1, at])
                pcb = torch.cat([pcb, torch.zeros_like(miss)])
                pout = torch.cat([pout, torch.full_like(miss, -1)])
                if pkh is not None:
                    pkh = torch.cat([pkh, at])
            # inside a stage: longest first -- over the whole block, or (row_bucket > 0) inside buckets of that many
            # consecutive rows taken in row order, so that rows with nearby ids run at about the same time: on a graph
            # with locality (communities of consecutive ids, what a clustering order at preparation time gives) they
            # share B rows, which then stay in L2
            bucket = int(row_bucket or 0)
            self.row_bucket = bucket
            nbk = -(-n // bucket) if bucket > 0 else 1
            pbk = prow // bucket if bucket > 0 else torch.zeros_like(prow)
            order = torch.argsort((pcb * nbk + pbk) * (piece_len + 1) + (piece_len - (pk1 - pk0)), stable=True)
            self.seg = torch.stack([prow, pk0, pk1, pout], dim=1)[order].int().contiguous()
            if pkh is not None:
                self.seg_hot = pkh[order].int().contiguous()
            self.stage_seg_off = torch.cat([torch.zeros(1, **i64), torch.cumsum(torch.bincount(pcb, minlength=nstage), 0)]).cpu().contiguous()
            cut_runs = torch.nonzero(is_cut).flatten()
            ccnt = pieces[cut_runs]
            cbase = torch.cumsum(ccnt, 0) - ccnt
            cut = torch.stack([run_row[cut_runs], cbase, ccnt, torch.zeros_like(ccnt)], dim=1)
            corder = torch.argsort(run_cb[cut_runs], stable=True)
            self.cut = cut[corder].int().contiguous()
            self.stage_cut_off = torch.cat([torch.zeros(1, **i64), torch.cumsum(torch.bincount(run_cb[cut_runs], minlength=nstage), 0)]).cpu().contiguous()
            self.npartial = int(ccnt.sum().item())
        self.nseg = self.seg.size(0)
        self.ncut = self.cut.size(0)
        self.c = _CPlan(nstage, self.seg.data_ptr(), self.stage_seg_off.data_ptr(), self.cut.data_ptr(),
                        self.stage_cut_off.data_ptr(), self.npartial, int(piece_len), covers)
        if self.hot_rows:
            self.c.hot_rows, self.c.hot_cols, self.c.seg_hot = self.hot_rows, self.hot_cols.data_ptr(), self.seg_hot.data_ptr()
        self._partial = {}

    def partial(self, ldb, dev):
        """Partial-row scratch of this plan, one buffer per stream (calls on one stream are ordered; two streams
        aggregating with the same block must not share piece rows)."""
        need = max(self.npartial, 1) * ldb
        key = torch.cuda.current_stream(dev).cuda_stream
        buf = self._partial.get(key)
        if buf is None or buf.numel() < need:
            buf = self._partial[key] = torch.empty(need, dtype=torch.float32, device=dev)
        return buf


class DevicePlan:
    """``struct dgnn_plan`` built by the library on the device (dgnn_plan_build): three small kernels over
    the rows instead of SegPlan's sorts over all nonzeros, no host synchronisation. The lists are sized by
    upper bounds, so ``nseg`` / ``ncut`` are bounds too. Same results as a SegPlan for every uncut run."""

    def __init__(self, A, nstage=None, piece_len=None, F=256):
        lay = _CLayout()
        piece = piece_len if piece_len is not None else int(os.environ.get('DGNN_PIECE_LEN', 0))
        _check(_lib.dgnn_plan_layout_for(A.n, A.m, A.nnz, padded_width(F), int(nstage or 0), int(piece or 0),
                                         ctypes.byref(lay)), 'plan_layout_for')
        self.layout = lay
        self.nstage, self.piece_len = lay.nstage, lay.piece
        if A.nnz == 0:                                   # nothing to list: an empty plan (beta scaling still runs)
            self.ws = torch.zeros(4, dtype=torch.int32, device=A.device)
            self.stage_seg_off = torch.zeros(lay.nstage + 1, dtype=torch.long)
            self.stage_cut_off = torch.zeros(lay.nstage + 1, dtype=torch.long)
            self.npartial = self.nseg = self.ncut = 0
            self.c = _CPlan(lay.nstage, self.ws.data_ptr(), self.stage_seg_off.data_ptr(), self.ws.data_ptr(),
                            self.stage_cut_off.data_ptr(), 0, lay.piece, 0)
            self._partial = {}
            return
        self.ws = torch.empty(lay.ws_ints, dtype=torch.int32, device=A.device)
        with torch.cuda.device(A.device):
            _check(_lib.dgnn_plan_build(A.rowptr.data_ptr(), A.col.data_ptr(), A.n, A.m, ctypes.byref(lay),
                                        self.ws.data_ptr(), _stream(A.val)), 'plan_build')
        steps = torch.arange(lay.nstage + 1, dtype=torch.long)
        self.stage_seg_off = (steps * lay.nseg_max).contiguous()
        self.stage_cut_off = (steps * lay.ncut_max).contiguous()
        self.npartial = lay.npartial_max
        self.nseg, self.ncut = lay.nstage * lay.nseg_max, lay.nstage * lay.ncut_max
        base = self.ws.data_ptr()
        self.c = _CPlan(lay.nstage, base + 4 * lay.hdr_ints, self.stage_seg_off.data_ptr(),
                        base + 4 * (lay.hdr_ints + 4 * lay.nstage * lay.nseg_max), self.stage_cut_off.data_ptr(),
                        self.npartial, lay.piece, 1)
        self._partial = {}

    def counts(self):
        """(segments, cut runs, partial rows) actually used -- reads the counters back (synchronises)."""
        if self.nseg == 0:
            return 0, 0, 0
        hdr = self.ws[:self.layout.hdr_ints].cpu().view(-1)
        ncls, per_stage = 256, 2 * 256 + 16                          # PLAN_NCLS, PLAN_HDR_STAGE in the library
        per = [hdr[st * per_stage:(st + 1) * per_stage] for st in range(self.nstage)]
        return (int(sum(int(p[:ncls].sum()) for p in per)), int(sum(int(p[2 * ncls]) for p in per)),
                int(hdr[16 * per_stage]))

    partial = SegPlan.partial


def strict():
    """DGNN_STRICT=1: parity / debugging runs. Row strips, dense tiles and row bundles are off and no run is cut into
    pieces, so EVERY row is one chain of fused multiply-adds in ascending column order -- the bits of the CPU oracle and
    of dgnn_spmm_csr_f32 on every row, and a non-finite feature reaches exactly the rows that reference it (dense tiles
    and bundles give absent entries a zero weight, which turns an Inf or NaN of a neighbour's neighbour into NaN;
    INTEGRATION.md section 4). Costs about 2x on the Reddit shape."""
    return os.environ.get('DGNN_STRICT', '0') == '1'


def _plan_for(A, F):
    """The resident plan of a block for feature width F, built once. Default: the host-side SegPlan, whose
    lists are exactly longest-first (1.7 % faster per epoch than the 12.5 %-wide length classes of a
    device-built plan, measured on Reddit and products shapes). DGNN_PLAN_BUILDER=device builds it with
    dgnn_plan_build instead: no sorts over the nonzeros, no host synchronisation, a third of the set-up
    time -- what very large blocks want."""
    # Dense tiles (DensePlan): where a block holds a large share of its nonzeros in tiles of 32 rows x their common
    # columns (hub rows x hub columns), that part runs as register-tiled fp32 FMAs and the plans below are built for the
    # sparse remainder. Feature blocks of at least 128 floats only: the tile kernel has a 64-float form too, but its cost
    # is dominated by the value stream, which does not shrink with the width (Reddit shape, F = 41: tiles 0.63 + remainder
    # 1.16 ms against 1.49 ms for the whole block on the stage kernel).
    if strict():
        plans = A.__dict__.setdefault('_plans', {})
        key = ('strict', default_stage_count(A.n, A.m, A.nnz, F))
        if key not in plans:
            plans[key] = SegPlan(A, key[1], piece_len=2 ** 31 - 1)
        return plans[key]
    if padded_width(F) >= 128 and not int(os.environ.get('DGNN_HOT_ROWS', 0) or 0) \
            and os.environ.get('DGNN_PLAN_BUILDER', 'torch') != 'device':
        # Row strips (StripPlan): tiles of 256 rows whose union column list is streamed once through LDS -- one B slice
        # per (tile, column) instead of one gather per nonzero. Takes precedence over dense tiles and bundles where the
        # rows share enough columns; what it leaves (the tail of short rows) goes through the plans below.
        S = _strip_for(A)
        if S is not None:
            hp = A.__dict__.setdefault('_strip_hybrid', {})
            rest_plan = _plan_for_sparse(S.rest, F)
            if id(rest_plan) not in hp:
                fresh = None
                if type(rest_plan) is SegPlan and not rest_plan.hot_rows and os.environ.get('DGNN_STRIP_FRESH', '1') == '1':
                    # the same runs, cut the same way, without the empty runs of the tiles' rows
                    fresh = SegPlan(S.rest, rest_plan.nstage, rest_plan.piece_len, row_bucket=rest_plan.row_bucket,
                                    skip_rows=S.strip_rows())
                hp[id(rest_plan)] = StripHybridPlan(S, rest_plan, fresh)
            return hp[id(rest_plan)]
        D = _dense_for(A)
        if D is not None:
            hp = A.__dict__.setdefault('_hybrid', {})
            rest_plan = _plan_for_sparse(D.rest, F)
            if id(rest_plan) not in hp:
                hp[id(rest_plan)] = HybridPlan(D, rest_plan)
            return hp[id(rest_plan)]
    return _plan_for_sparse(A, F)


def _plan_for_sparse(A, F):
    ns = default_stage_count(A.n, A.m, A.nnz, F)
    # DGNN_HOT_ROWS=256|512 (opt-in, measured slower on every shape tried): hot-tile plans -- the
    # B rows of each stage's most referenced columns staged in LDS. Only for blocks with enough work per stage.
    hot = int(os.environ.get('DGNN_HOT_ROWS', 0) or 0)
    if hot and A.nnz < (1 << 22):
        hot = 0
    plans = A.__dict__.setdefault('_plans', {})
    # Row bundles (BundlePlan): 4 rows share one gather per union column. Worth it where the rows of a block share
    # many columns -- decided per block by building the plan once and reading its union ratio (gathers per nonzero;
    # Reddit shape 0.74: -13..-20 % time; sparse shapes ~1: more index bytes for nothing) -- and only for feature
    # blocks of at least 128 floats (narrower ones are not bound by the gathers). DGNN_BUNDLE=0 turns them off,
    # DGNN_BUNDLE=1 forces them.
    want = os.environ.get('DGNN_BUNDLE', 'auto')
    bundle = (not hot and want != '0' and padded_width(F) >= 128 and A.nnz >= (1 << 16)
              and os.environ.get('DGNN_PLAN_BUILDER', 'torch') != 'device')
    if bundle and want != '1':
        ok = A.__dict__.get('_bundle_ok')
        if ok is None:                                          # one cheap test per block: distinct (row pair, column)
            ok = A.__dict__['_bundle_ok'] = _bundle_worth_it(A, limit=A.__dict__.get('_bundle_limit', 0.85))
        bundle = ok
    # Row-bucketed work order for blocks with locality (rows with nearby ids reference the same B rows: a graph with
    # communities after a clustering order at preparation time): runs are taken longest first inside buckets of
    # ROW_BUCKET consecutive rows instead of over the whole block, so the rows in flight share B rows in L2, and the
    # column-block stages (a substitute for that locality) are dropped. Measured on a products-sized graph with planted
    # communities: 17.8 -> 9.0 ms at F = 256; on the scrambled R-MAT shapes (no locality) the same order costs 30 %, so
    # the rule looks at the block: at least half of a row bucket's nonzeros in ONE column bucket. DGNN_ROW_BUCKET=0 / n
    # overrides.
    bucket = 0
    if not bundle and not hot and os.environ.get('DGNN_PLAN_BUILDER', 'torch') != 'device':
        env_b = os.environ.get('DGNN_ROW_BUCKET')
        if env_b is not None and env_b != '':
            bucket = int(env_b)
        elif A.nnz >= (1 << 20) and locality_share(A) >= 0.5:
            bucket = ROW_BUCKET
    if bucket:
        ns = 1
    key = (ns, hot, 'bundle' if bundle else 'runs', bucket)
    if key not in plans:
        if bundle:
            plans[key] = BundlePlan(A, bundle_stage_count(A.n, A.m, A.nnz, F), 4)
        elif hot:
            plans[key] = SegPlan(A, ns, hot_rows=hot)
        elif os.environ.get('DGNN_PLAN_BUILDER', 'torch') == 'device':
            plans[key] = DevicePlan(A, ns, None, F)
        else:
            plans[key] = SegPlan(A, ns, row_bucket=bucket)
    return plans[key]


ROW_BUCKET = 4096


def locality_share(A, bucket=ROW_BUCKET):
    """Share of the block's nonzeros that sit, for their bucket of `bucket` consecutive rows, in that bucket's most
    referenced bucket of `bucket` consecutive columns (1 = block diagonal at that granularity, ~1 / column buckets for a
    scrambled graph). One bincount over the nonzeros, cached on the block."""
    got = A.__dict__.get('_locality')
    if got is None or got[0] != bucket:
        if A.nnz == 0:
            share = 0.0
        else:
            rp = A.rowptr.long()
            rb = torch.repeat_interleave(torch.arange(A.n, device=A.device) // bucket, rp[1:] - rp[:-1])
            ncb = -(-A.m // bucket)
            cnt = torch.bincount(rb * ncb + A.col.long() // bucket, minlength=(-(-A.n // bucket)) * ncb)
            share = float(cnt.view(-1, ncb).max(dim=1).values.sum().item()) / A.nnz
        got = A.__dict__['_locality'] = (bucket, share)
    return got[1]


def bundle_stage_count(n, m, nnz, F):
    """Column-block stages of a bundle plan: a bundle's run is the union of four rows' runs, so runs stay long with more
    stages than default_stage_count allows (measured on the Reddit shape: 6 instead of 4 at 602 floats)."""
    if n == 0 or nnz == 0:
        return 1
    by_cache = -(-m * 256 // (10 << 20))
    by_density = max(1, int(4 * nnz / max(n, 1) / 24))
    return int(max(1, min(by_cache, by_density, 16)))


def _bundle_worth_it(A, R=4, limit=0.85):
    """True when tying R rows of similar length together cuts the gathers to at most ``limit`` of the nonzeros:
    rows ranked by length, R consecutive ones a group, distinct (group, column) pairs counted (one stage)."""
    if A.nnz == 0 or A.n < 2 * R:
        return False
    rp = A.rowptr.long()
    deg = rp[1:] - rp[:-1]
    if float(A.nnz) / A.n < 16:                                 # a few nonzeros per row: nothing to share
        return False
    order = torch.argsort(deg, descending=True, stable=True)
    rank = torch.empty_like(order)
    rank[order] = torch.arange(A.n, device=A.device)
    grp = torch.repeat_interleave(rank // R, deg)
    distinct = torch.unique(grp * A.m + A.col.long()).numel()
    return distinct <= limit * A.nnz


def spmm_plan(A, B, C, alpha=1.0, beta=1.0, plan=None, overwrite=False):
    """C <- alpha * A @ B[:, :F] + beta * C, fast path. ``B`` is a padded feature block from
    ``pad_rows`` (16-byte aligned rows); ``C`` is any row-major [n, F] fp32 matrix.
    ``overwrite``: C <- alpha * A @ B whatever C holds (it may be uninitialised); beta is ignored."""
    bptr, ldb = _dense(B, 'B')
    cptr, ldc = _dense(C, 'C')
    F = C.size(1)
    if B.size(0) != A.m or C.size(0) != A.n or B.size(1) < F:
        raise RuntimeError(f'spmm_plan: shape mismatch A[{A.n}x{A.m}] B{tuple(B.shape)} C{tuple(C.shape)}')
    if not (A.device == B.device == C.device):
        raise RuntimeError('spmm_plan: operands are on different devices')
    if plan is None:
        plan = _plan_for(A, F)
    if isinstance(plan, StripHybridPlan):                # the rest first (it owns beta / overwrite and lists every row), then the strips
        if float(alpha) != 1.0:
            raise RuntimeError('spmm_plan: row-strip plans multiply with alpha == 1 only')
        if overwrite and plan.rest_fresh is not None:     # every row written once: the rest's rows by its plan, the tiles' by the strips
            spmm_plan(plan.strip.rest, B, C, alpha, beta, plan=plan.rest_fresh, overwrite=True)
            spmm_strip(plan.strip, B, C, fresh=True)
            return
        spmm_plan(plan.strip.rest, B, C, alpha, beta, plan=plan.rest_plan, overwrite=overwrite)
        spmm_strip(plan.strip, B, C)
        return
    if isinstance(plan, HybridPlan):                     # sparse remainder first (it owns beta / overwrite), then the tiles
        spmm_plan(plan.dense.rest, B, C, alpha, beta, plan=plan.rest_plan, overwrite=overwrite)
        spmm_dense(plan.dense, B, C, alpha)
        return
    dev = C.device
    P = plan.partial(ldb, dev)
    if isinstance(plan, BundlePlan):
        fn = _lib.dgnn_spmm_bundle_f32
        args = (ctypes.byref(plan.c), A.n, A.m, bptr, ldb, cptr, ldc, P.data_ptr(), ldb, F, float(alpha),
                0.0 if overwrite else float(beta))
    else:
        col, val = getattr(plan, 'col', A.col), getattr(plan, 'val', A.val)      # a hot-tile plan indexes its own entry arrays
        args = (ctypes.byref(plan.c), col.data_ptr(), val.data_ptr(), A.n, A.m, bptr, ldb, cptr, ldc,
                P.data_ptr(), ldb, F, float(alpha))
        if overwrite:
            fn = _lib.dgnn_spmm_plan_overwrite_f32
        else:
            fn, args = _lib.dgnn_spmm_plan_f32, args + (float(beta),)
    if torch.cuda.current_device() == dev.index:         # the usual case: no device switch to pay for
        rc = fn(*args, _stream(C))
    else:
        with torch.cuda.device(dev):
            rc = fn(*args, _stream(C))
    _check(rc, 'spmm_plan')


_pad_cache = {}


def _pad_buffer(n, ld, dev):
    """One reusable padded block per (shape, device, stream): calls on one stream are ordered."""
    key = (n, ld, dev, torch.cuda.current_stream(dev).cuda_stream)
    buf = _pad_cache.get(key)
    if buf is None:
        if len(_pad_cache) > 64:
            _pad_cache.clear()
        buf = _pad_cache[key] = torch.empty(n, ld, dtype=torch.float32, device=dev)
    return buf


def aggregate(A, H, Z, overwrite=False):
    """Z += A @ H for a CsrBlock A: the SpMM of one pipeline stage (models/gcn.py:28-29).
    H is any row-major [m, F] fp32 matrix; unless its rows already are 128-byte aligned it is
    first copied into a padded block (dgnn_pad_rows_f32). ``overwrite``: Z = A @ H instead (the first
    stage of an aggregation: Z may be uninitialised, it is written without being read)."""
    F = Z.size(1)
    if not (H.dim() == 2 and H.size(1) == F):
        raise RuntimeError(f'aggregate: H{tuple(H.shape)} does not match Z{tuple(Z.shape)}')
    if H.stride(0) != padded_width(F) or H.data_ptr() % 128 != 0 or H.stride(1) != 1:
        with torch.cuda.device(H.device):
            H = pad_rows(H, _pad_buffer(H.size(0), padded_width(F), H.device))
    spmm_plan(A, H, Z, 1.0, 1.0, overwrite=overwrite)


# --------------------------------------------------------------------------- row bundles (register reuse across rows)
class BundlePlan:
    """``struct dgnn_bundle_plan`` for one CsrBlock: R rows that share columns are multiplied together
    (``spmm_bundle_kernel``). Per stage the rows are taken longest first, R at a time; a bundle walks the union of its
    rows' columns once and every gathered slice of B feeds up to R accumulators. Per union entry: the column and R
    values (0 where a row lacks it). Every row still sums its own nonzeros in ascending column order, so rows of uncut
    bundles carry the strict chain's bits (``uncut_rows``). Integer work with torch ops on the block's device, once."""

    def __init__(self, A, nstage=1, R=4, piece_len=None):
        if R not in (2, 4, 6, 8):
            raise ValueError('R must be 2, 4, 6 or 8')
        A.require_distinct('BundlePlan')
        dev = A.device
        n, m, nnz = A.n, A.m, A.nnz
        nstage = max(1, int(nstage))
        if piece_len is None:
            piece_len = int(os.environ.get('DGNN_PIECE_LEN', 0)) or default_piece_len(nnz, nstage)
        self.nstage, self.R, self.piece_len = nstage, R, int(piece_len)
        i64 = dict(dtype=torch.long, device=dev)
        rp = A.rowptr.long()
        cbw = max(1, -(-m // nstage))
        row_idx = torch.repeat_interleave(torch.arange(n, **i64), rp[1:] - rp[:-1])
        col = A.col.long()
        cb = col // cbw
        seglen = torch.bincount(row_idx * nstage + cb, minlength=n * nstage).view(n, nstage)
        NB = max(1, -(-n // R))
        order = torch.argsort(seglen, dim=0, descending=True, stable=True)            # order[i, s]: i-th longest row of stage s
        rank = torch.empty_like(order)
        rank.scatter_(0, order, torch.arange(n, **i64)[:, None].expand(n, nstage).contiguous())
        rnk = rank[row_idx, cb]
        gb = cb * NB + rnk // R                                                       # global bundle id of every nonzero
        ukey, inv = torch.unique(gb * m + col, return_inverse=True)                   # sorted: by bundle, then column
        nu = int(ukey.numel())
        ub = ukey // m
        self.ucol = (ukey % m).int().contiguous() if nu else torch.zeros(1, dtype=torch.int32, device=dev)   # never a null pointer
        del ukey
        uval = torch.zeros(max(nu, 1) * R, dtype=torch.float32, device=dev)
        uval[inv * R + rnk % R] = A.val
        self.uval = uval
        self.union_ratio = nu / max(nnz, 1)                                           # gathers per nonzero
        cnt_u = torch.bincount(ub, minlength=nstage * NB)                             # union size of every bundle
        k1 = torch.cumsum(cnt_u, 0)
        k0 = k1 - cnt_u
        del ub, inv, rnk, gb, row_idx, col, cb
        # bundles that exist: stage 0 lists every row (a product that overwrites C stores them all), the other stages
        # the rows with a run there
        rows_in = torch.cat([torch.tensor([n], **i64), (seglen[:, 1:] > 0).sum(0)])
        nb_s = -(-rows_in // R)
        stage_of = torch.repeat_interleave(torch.arange(nstage, **i64), nb_s)
        b_in = torch.arange(int(nb_s.sum()), **i64) - (torch.cumsum(nb_s, 0) - nb_s)[stage_of]
        gid = stage_of * NB + b_in
        midx = b_in[:, None] * R + torch.arange(R, **i64)[None, :]
        valid = midx < rows_in[stage_of][:, None]
        members = torch.where(valid, order[midx.clamp(max=max(n - 1, 0)), stage_of[:, None].expand(-1, R)],
                              torch.full_like(midx, -1))
        del order, rank, seglen
        length = cnt_u[gid]
        pieces = torch.clamp((length + piece_len - 1) // piece_len, min=1)
        of = torch.repeat_interleave(torch.arange(gid.numel(), **i64), pieces)
        first = torch.cumsum(pieces, 0) - pieces
        pidx = torch.arange(of.numel(), **i64) - first[of]
        pk0 = k0[gid][of] + pidx * piece_len
        pk1 = torch.minimum(pk0 + piece_len, k1[gid][of])
        is_cut = pieces > 1
        pcut = is_cut[of]
        pout = torch.where(pcut, (torch.cumsum(pcut.long(), 0) - 1) * R, torch.full_like(pk0, -1))
        pstage = stage_of[of]
        o2 = torch.argsort(pstage * (piece_len + 1) + (piece_len - (pk1 - pk0)), stable=True)
        desc = torch.cat([members[of], pk0[:, None], pk1[:, None], pout[:, None], torch.zeros_like(pk0)[:, None]], dim=1)
        self.desc = desc[o2].int().contiguous()
        self.stage_desc_off = torch.cat([torch.zeros(1, **i64), torch.cumsum(torch.bincount(pstage, minlength=nstage), 0)]).cpu().contiguous()
        self.ndesc = int(self.desc.size(0))
        self.npartial = int(pcut.sum().item()) * R
        # cut list: one entry per (cut bundle, member row): {row, first partial row, pieces, R - 1}
        cb_idx = torch.nonzero(is_cut).flatten()
        first_out = torch.zeros(gid.numel(), **i64)
        head = pcut & (pidx == 0)
        first_out[of[head]] = pout[head]
        crow = members[cb_idx]
        cbase = first_out[cb_idx][:, None] + torch.arange(R, **i64)[None, :]
        ccnt = pieces[cb_idx][:, None].expand(-1, R)
        cst = stage_of[cb_idx][:, None].expand(-1, R)
        ok = crow >= 0
        cut = torch.stack([crow[ok], cbase[ok], ccnt[ok], torch.full_like(crow[ok], R - 1)], dim=1)
        co = torch.argsort(cst[ok], stable=True)
        self.cut = cut[co].int().contiguous()
        self.stage_cut_off = torch.cat([torch.zeros(1, **i64), torch.cumsum(torch.bincount(cst[ok], minlength=nstage), 0)]).cpu().contiguous()
        self.ncut = int(self.cut.size(0))
        self._uncut = torch.ones(n, dtype=torch.bool, device=dev)
        self._uncut[crow[ok]] = False
        self.nbytes = 4 * (self.desc.numel() + self.ucol.numel() + self.uval.numel())
        self.c = _CBundlePlan(nstage, R, self.desc.data_ptr(), self.stage_desc_off.data_ptr(), self.ucol.data_ptr(),
                              self.uval.data_ptr(), self.cut.data_ptr(), self.stage_cut_off.data_ptr(), self.npartial)
        self._partial = {}

    partial = SegPlan.partial

    def uncut_rows(self):
        """Rows whose result is one strict chain (no bundle they belong to is cut in any stage)."""
        return self._uncut


def _segplan_uncut_rows(self):
    """Rows no stage cuts into pieces: their result is the strict chain over the plan's entry order."""
    m = torch.ones(self.n, dtype=torch.bool, device=self.seg.device)
    if self.cut.numel():
        m[self.cut[:, 0].long()] = False
    return m


SegPlan.uncut_rows = _segplan_uncut_rows


# --------------------------------------------------------------------------- dense tiles (the part of a block that is not sparse)
DENSE_TR = 32                # rows of a dense tile (DENSE_TR in the library)


class DensePlan:
    """``struct dgnn_dense_plan`` for one CsrBlock, plus the sparse remainder of the block.

    Rows are ranked by length and taken ``DENSE_TR`` = 32 at a time (a tile). A column that at least ``min_count`` of a
    tile's rows reference is *listed* for that tile; tiles with fewer than ``min_cols`` listed columns list nothing
    (their rows stay untouched: the strict chain). The listed entries leave the CSR block -- ``self.rest`` is the
    CsrBlock of what remains -- and are stored densely: ``kval[32 * k + r]`` = value of member r in listed column k
    (0 where the row lacks it). ``spmm_dense_kernel`` walks a tile's list once per 128-float feature chunk: one
    coalesced slice of a B row per listed column feeds 32 rows. A list much longer than ``piece_len`` is cut into pieces
    at column boundaries (partial rows, merged in piece order). Integer work with torch ops on the block's device, once
    per block. Defaults from sweeps on the Reddit shape (profiles/r02_dense_*.txt): min_count 5, min_cols 128,
    piece_len 1024.

    ``share``: fraction of the block's nonzeros that moved into tiles; ``density``: nonzeros per tile cell.
    """

    def __init__(self, A, min_count=None, min_cols=None, piece_len=None, keep_mask=False):
        TR = DENSE_TR
        A.require_distinct('DensePlan')
        dev = A.device
        n, m, nnz = A.n, A.m, A.nnz
        if min_count is None:
            min_count = int(os.environ.get('DGNN_DENSE_MIN_COUNT', 5))
        if min_cols is None:
            min_cols = int(os.environ.get('DGNN_DENSE_MIN_COLS', 128))
        if piece_len is None:
            piece_len = int(os.environ.get('DGNN_DENSE_PIECE', 1024))
        if piece_len < 8 or piece_len % 8:
            raise ValueError('piece_len must be a multiple of 8')
        self.min_count, self.min_cols, self.piece_len = int(min_count), int(min_cols), int(piece_len)
        i64 = dict(dtype=torch.long, device=dev)
        rp = A.rowptr.long()
        deg = rp[1:] - rp[:-1]
        ntile = max(1, -(-n // TR))
        order = torch.argsort(deg, descending=True, stable=True)
        rank = torch.empty_like(order)
        rank[order] = torch.arange(n, **i64)
        row_idx = torch.repeat_interleave(torch.arange(n, **i64), deg)
        rnk = rank[row_idx]
        col = A.col.long()
        ukey, inv, cnt = torch.unique((rnk // TR) * m + col, return_inverse=True, return_counts=True)   # by tile, then column
        utile = ukey // m
        sel = cnt >= self.min_count
        per_tile = torch.bincount(utile[sel], minlength=ntile)
        sel &= (per_tile >= self.min_cols)[utile]                       # tiles with a short list keep their rows whole
        # A tile's list is cut into pieces at COLUMN boundaries: a tile with many listed columns splits the column range
        # into nb (a power of two, ~piece_len listed columns per part) equal blocks, and the pieces of all tiles are
        # worked off in column order -- the waves in flight at any moment read B rows of one narrow column range, which
        # the L2s hold (measured: the tiles' B traffic is what bounds the kernel otherwise). A piece's slots are padded
        # to a multiple of 8 columns (what the kernel takes per loop trip): pad slots repeat its last column with zero
        # weights.
        lt = utile[sel]
        lc = ukey[sel] % m
        listed = int(lt.numel())
        cnt_t = torch.bincount(lt, minlength=ntile)
        want = torch.clamp((cnt_t + self.piece_len - 1) // self.piece_len, min=1)
        nb_t = torch.ones_like(want)
        while bool((nb_t < want).any()):
            nb_t = torch.where(nb_t < want, nb_t * 2, nb_t)
        bw_t = (m + nb_t - 1) // nb_t
        pbase_t = torch.cumsum(nb_t, 0) - nb_t
        npot = int(nb_t.sum().item())
        pid = pbase_t[lt] + lc // bw_t[lt]                              # ascending: the pairs are sorted by (tile, column)
        cnt_piece = torch.bincount(pid, minlength=npot)
        cnt_pp = (cnt_piece + 7) // 8 * 8
        k1 = torch.cumsum(cnt_pp, 0)
        k0 = k1 - cnt_pp
        nk = int(k1[-1].item()) if npot else 0
        first_of_piece = torch.cumsum(cnt_piece, 0) - cnt_piece
        slot_l = k0[pid] + (torch.arange(listed, **i64) - first_of_piece[pid])
        kcol = torch.zeros(max(nk, 1), **i64)
        kcol[slot_l] = lc
        for j in range(1, 8):
            short = torch.nonzero(((cnt_pp - cnt_piece) >= j) & (cnt_piece > 0)).flatten()
            kcol[k1[short] - j] = kcol[k0[short] + cnt_piece[short] - 1]
        self.kcol = kcol.int().contiguous()
        slot = torch.full((ukey.numel(),), -1, **i64)
        slot[sel] = slot_l
        in_dense = sel[inv]
        kval = torch.zeros(max(nk, 1) * TR, dtype=torch.float32, device=dev)
        kval[slot[inv[in_dense]] * TR + rnk[in_dense] % TR] = A.val[in_dense]
        self.kval = kval
        self.nk, self.listed = nk, listed
        moved = int(in_dense.sum().item())
        self.share = moved / max(nnz, 1)
        self.density = moved / max(listed * TR, 1)
        self.in_dense = in_dense if keep_mask else None                 # tests: which entries of A moved
        # the sparse remainder, as a CsrBlock of its own
        keep = ~in_dense
        rest_rp = torch.zeros(n + 1, dtype=torch.int32, device=dev)
        rest_rp[1:] = torch.cumsum(torch.bincount(row_idx[keep], minlength=n), 0).int()
        self.rest = CsrBlock(n, m, rest_rp, A.col[keep].contiguous(), A.val[keep].contiguous())
        # what is left of a block whose hub core went into tiles still shares columns (Reddit shape: 0.956 gathers per
        # nonzero at R = 4); measured with bundles: F = 602 the same time as runs, F = 256 4 % less (profiles/r02_dense8*.txt)
        self.rest.__dict__['_bundle_limit'] = 0.97
        del ukey, inv, cnt, utile, sel, slot, slot_l, pid, lt, lc, in_dense, keep, row_idx, rnk, col
        # pieces in working order: by first column of their column block, longest first inside one
        rows = torch.full((ntile * TR,), -1, **i64)
        rows[:n] = order
        self.rows = rows.int().contiguous()
        pidx = torch.nonzero(cnt_piece > 0).flatten()                  # in (tile, block) order
        ptile = torch.repeat_interleave(torch.arange(ntile, **i64), nb_t)[pidx]
        pstart = (pidx - pbase_t[ptile]) * bw_t[ptile]
        plen = cnt_pp[pidx]
        per_tile_pieces = torch.bincount(ptile, minlength=ntile)
        tid = torch.nonzero(per_tile_pieces > 0).flatten()
        self.tiles = tid                                               # tiles that list something
        pcut = (per_tile_pieces > 1)[ptile]
        pout = torch.where(pcut, (torch.cumsum(pcut.long(), 0) - 1) * TR, torch.full_like(plen, -1))
        maxlen = int(plen.max().item()) if plen.numel() else 0
        self.max_piece = maxlen
        # whole (uncut) lists span every column and have no place in that order: they go last, where they fill the tail of
        # the launch (measured against first / centre / end of their range: tiles -2 % at F = 602, -4 % at F = 256)
        pkey = torch.where(nb_t[ptile] == 1, torch.full_like(pstart, m), pstart)
        o2 = torch.argsort(pkey * (maxlen + 1) + (maxlen - plen), stable=True)
        self.desc = torch.stack([k0[pidx], k1[pidx], ptile, pout], dim=1)[o2].int().contiguous()
        self.npiece = int(self.desc.size(0))
        self.npartial = int(pcut.sum().item()) * TR
        # cut list: one entry per (cut tile, member row): {row, first partial row, pieces, TR - 1}
        ct = torch.nonzero(per_tile_pieces > 1).flatten()              # cut tiles
        first_piece_of_tile = torch.cumsum(per_tile_pieces, 0) - per_tile_pieces
        first_out = pout[first_piece_of_tile[ct]] if ct.numel() else torch.zeros(0, **i64)
        crow = self.rows.long().view(ntile, TR)[ct]
        cbase = first_out[:, None] + torch.arange(TR, **i64)[None, :]
        ccnt = per_tile_pieces[ct][:, None].expand(-1, TR)
        ok = crow >= 0
        self.cut = torch.stack([crow[ok], cbase[ok], ccnt[ok], torch.full_like(crow[ok], TR - 1)], dim=1).int().contiguous()
        self.ncut = int(self.cut.size(0))
        self.ntile = ntile
        touched = torch.zeros(ntile * TR, dtype=torch.bool, device=dev)
        touched.view(ntile, TR)[tid] = True
        self._touched_rows = torch.zeros(n, dtype=torch.bool, device=dev)
        self._touched_rows[order] = touched[:n]
        self._cut_rows = torch.zeros(n, dtype=torch.bool, device=dev)
        self._cut_rows[crow[ok]] = True
        self.nbytes = 4 * (self.desc.numel() + self.rows.numel() + self.kcol.numel() + self.kval.numel() + self.cut.numel())
        self.c = _CDensePlan(TR, ntile, self.npiece, self.ncut, self.desc.data_ptr(), self.rows.data_ptr(),
                             self.kcol.data_ptr(), self.kval.data_ptr(), self.cut.data_ptr(), nk, self.npartial)
        self._partial = {}

    partial = SegPlan.partial

    def touched_rows(self):
        """Rows that belong to a tile with a list: their chain is [remainder entries, then the listed entries]."""
        return self._touched_rows

    def cut_rows(self):
        """Rows of tiles whose list is cut into pieces (fixed-order piece sums instead of one chain)."""
        return self._cut_rows


class HybridPlan:
    """A block as dense tiles + sparse remainder: ``rest_plan`` (a SegPlan / BundlePlan of ``dense.rest``) runs first,
    with the call's beta / overwrite; ``dgnn_spmm_dense_f32`` then adds the tiles."""

    def __init__(self, dense, rest_plan):
        self.dense, self.rest_plan = dense, rest_plan

    def uncut_rows(self):
        """Rows whose result is one strict chain over the plan's own entry order (remainder ascending, then listed
        columns ascending)."""
        return self.rest_plan.uncut_rows() & ~self.dense.cut_rows()

    def strict_rows(self):
        """Rows whose result is the strict chain over A's own (ascending column) order: uncut and in no tile."""
        return self.rest_plan.uncut_rows() & ~self.dense.touched_rows()


def spmm_dense(D, B, C, alpha=1.0):
    """C += alpha * (dense tiles of D) @ B[:, :F]; B a padded feature block (pad_rows)."""
    bptr, ldb = _dense(B, 'B')
    cptr, ldc = _dense(C, 'C')
    F = C.size(1)
    dev = C.device
    P = D.partial(ldb, dev)
    args = (ctypes.byref(D.c), D.rest.n, D.rest.m, bptr, ldb, cptr, ldc, P.data_ptr(), ldb, F, float(alpha))
    if torch.cuda.current_device() == dev.index:
        rc = _lib.dgnn_spmm_dense_f32(*args, _stream(C))
    else:
        with torch.cuda.device(dev):
            rc = _lib.dgnn_spmm_dense_f32(*args, _stream(C))
    _check(rc, 'spmm_dense')


# --------------------------------------------------------------------------- row strips (B slices staged in LDS)
STRIP_R, STRIP_NW, STRIP_W, STRIP_LINE = 256, 16, 64, 32
_STRIP_BITMAP_LIMIT = 2 ** 31          # (tile, column) cells up to which a tile's union comes from a bitmap (tests lower it: the sort path)     # rows per tile, waves per workgroup, widest strip, entries per line (the kernel's geometry)


class StripPlan:
    """``struct dgnn_strip_plan`` for one CsrBlock, plus what is left of the block (``self.rest``).

    Rows are ranked by length and taken 256 at a time (a tile: rows of similar length share the hub columns). A tile
    whose union of columns is at most ``max_ratio`` of its nonzeros is *selected*: its entries leave the CSR block and
    are laid out for ``spmm_strip_kernel`` -- the tile's union column list (``ucol``) is walked in strips of W columns
    whose B slices go through LDS once (W = 64, or 32 / 16 / 8 for tiles whose rows are so long that a wave's share of a
    64-column strip would not fit one line); wave w of the workgroup owns members w, w + 16, ... of the tile and reads,
    per strip, one block of (slot, row, value) entries sorted by row then column, stored as whole lines of 32 entries
    (``ent``; a line's first entry carries how many groups of 4 to take and whether the block ends with it). Tiles with
    poorer reuse (the tail of short rows) stay in ``self.rest``, a CsrBlock for the other plans. A union list longer than
    ``piece_strips`` strips is cut into pieces (partial rows, merged in piece order). Integer work with torch ops on the
    block's device, once per block.

    ``share``: fraction of the block's nonzeros in selected tiles; ``union_ratio``: B slices fetched per nonzero there;
    ``padding``: entry slots streamed per nonzero.
    """

    def __init__(self, A, max_ratio=None, piece_strips=None, min_tile_nnz=None, block_target=None):
        R, NW, WMAX, LINE = STRIP_R, STRIP_NW, STRIP_W, STRIP_LINE
        RW = R // NW
        A.require_distinct('StripPlan')
        dev = A.device
        n, m, nnz = A.n, A.m, A.nnz
        if max_ratio is None:
            max_ratio = float(os.environ.get('DGNN_STRIP_MAX_RATIO', 0.7))
        if piece_strips is None:
            piece_strips = int(os.environ.get('DGNN_STRIP_PIECE', 0))      # 0: by the number of strips, below
        if min_tile_nnz is None:
            min_tile_nnz = int(os.environ.get('DGNN_STRIP_MIN_TILE_NNZ', 1024))
        if block_target is None:
            block_target = float(os.environ.get('DGNN_STRIP_BLOCK', 52))
        if piece_strips < 0:
            raise ValueError('piece_strips must not be negative')
        self.max_ratio = float(max_ratio)
        i64 = dict(dtype=torch.long, device=dev)
        i32 = dict(dtype=torch.int32, device=dev)
        rp = A.rowptr.long()
        deg = rp[1:] - rp[:-1]
        ntile = max(1, -(-n // R))
        order = torch.argsort(deg, descending=True, stable=True)            # order[j] = the row of rank j
        tile_of_row = torch.empty(n, **i64)
        tile_of_row[order] = torch.arange(n, **i64) // R
        # The entries in RANK order (row of rank 0 first), columns ascending inside a row -- a gather, not a sort. Everything
        # per entry below is int32 and lives in this order: inside a (wave, strip) block it is already the order the kernel
        # wants (row, then column), so the one sort left is a stable sort by block number.
        deg_r = deg[order]
        j_e = torch.repeat_interleave(torch.arange(n, **i32), deg_r)        # rank of the entry's row
        start_r = torch.cumsum(deg_r, 0) - deg_r
        src = (rp[order] - start_r).int()[j_e] + torch.arange(nnz, **i32)   # where the entry sits in the CSR arrays
        del start_r
        col_r = A.col[src]
        tile_e = j_e // R
        ssz = torch.bincount(tile_e, minlength=ntile)
        bitmap = ntile * m < _STRIP_BITMAP_LIMIT
        if bitmap:
            # union of a tile's columns through a (tile, column) bitmap: no sort over the nonzeros
            key = tile_e * m + col_r
            present = torch.zeros(ntile * m, dtype=torch.bool, device=dev)
            present[key] = True
            pm = present.view(ntile, m)
            usz = pm.sum(1)
            slot = torch.cumsum(pm, 1, dtype=torch.int32).view(-1)          # 1 + position inside the tile's sorted union
            e_uslot = slot[key] - 1
            del slot, key
            ukey = None
        else:
            key = tile_e.long() * m + col_r
            ukey, inv = torch.unique(key, return_inverse=True)              # sorted: by tile, then column
            del key
            utile = ukey // m
            usz = torch.bincount(utile, minlength=ntile)
            e_uslot = (inv - (torch.cumsum(usz, 0) - usz)[tile_e]).int()
            del inv
        del col_r
        ok_t = (ssz >= max(int(min_tile_nnz), 1)) & (usz.double() <= self.max_ratio * ssz.double())
        # what stays: a CsrBlock of its own (whole rows: a tile is taken or left as a whole), in the block's own order
        keep_row = ~ok_t[tile_of_row]
        keep = torch.repeat_interleave(keep_row, deg)
        rest_rp = torch.zeros(n + 1, dtype=torch.int32, device=dev)
        rest_rp[1:] = torch.cumsum(deg * keep_row, 0).int()
        self.rest = CsrBlock(n, m, rest_rp, A.col[keep].contiguous(), A.val[keep].contiguous())
        self.rest._distinct = True
        del keep, keep_row
        sel_e = ok_t[tile_e]
        moved = int(sel_e.sum().item())
        self.share = moved / max(nnz, 1)
        # the selected tiles' union lists, compacted
        usz_s = torch.where(ok_t, usz, torch.zeros_like(usz))
        first = torch.cumsum(usz_s, 0) - usz_s
        nu = int(usz_s.sum().item())
        if not moved:
            self.ucol = torch.zeros(1, dtype=torch.int32, device=dev)
        elif ukey is None:
            self.ucol = torch.nonzero(pm & ok_t[:, None])[:, 1].int().contiguous()      # row-major: by tile, then column
        else:
            self.ucol = (ukey % m)[ok_t[utile]].int().contiguous()
            del ukey, utile
        if bitmap:
            del present, pm
        self.nu = nu
        self.union_ratio = nu / max(moved, 1)
        e_tile = tile_e[sel_e]
        e_uslot = e_uslot[sel_e]
        e_i = j_e[sel_e] % R
        e_val = A.val[src[sel_e]]
        del tile_e, j_e, src, sel_e
        # strip width of every tile: the widest of 64 / 32 / 16 / 8 that keeps a wave's share of a strip near one line
        per_col = ssz.double() / usz.clamp(min=1).double() / NW                 # entries per wave and union column
        w_t = torch.full((ntile,), WMAX, **i64)
        for w in (WMAX // 2, WMAX // 4, WMAX // 8):
            w_t = torch.where(per_col * (2 * w) > float(block_target), torch.full_like(w_t, w), w_t)
        w_e = w_t.int()[e_tile]
        e_strip, e_slot = e_uslot // w_e, e_uslot % w_e
        e_wave, e_rsub = e_i % NW, e_i // NW
        del e_uslot, e_i, w_e
        # strips and pieces of every tile
        nstrip_t = (usz_s + w_t - 1) // w_t
        # Piece length: one workgroup walks one piece per feature chunk, so a launch wants several pieces per CU whatever
        # the block's size -- the row slab of one of 8 ranks has an eighth of the tiles but hub tiles as long as the whole
        # graph's. Rule: total strips / 384, to the nearest power of two in 32 .. 512 (Reddit shape: 512 for the whole block,
        # 128 for a rank's slab at 8 ranks, where 512 costs 3x at F = 256: profiles/r03_slab_piece_sweep.txt).
        self.nstrips = int(nstrip_t.sum().item())
        if not piece_strips:
            want = max(self.nstrips / 384.0, 1.0)
            piece_strips = min(512, max(32, 1 << int(round(math.log2(want)))))
        self.piece_strips = int(piece_strips)
        np_t = (nstrip_t + self.piece_strips - 1) // self.piece_strips
        spp_t = torch.where(np_t > 0, (nstrip_t + np_t.clamp(min=1) - 1) // np_t.clamp(min=1), torch.zeros_like(np_t))
        np_t = torch.where(spp_t > 0, (nstrip_t + spp_t.clamp(min=1) - 1) // spp_t.clamp(min=1), torch.zeros_like(np_t))
        sbase_t = torch.cumsum(nstrip_t, 0) - nstrip_t
        pbase_t = torch.cumsum(np_t, 0) - np_t
        npiece = int(np_t.sum().item())
        nstrips = int(nstrip_t.sum().item())
        ptile = torch.repeat_interleave(torch.arange(ntile, **i64), np_t)
        pq = torch.arange(npiece, **i64) - pbase_t[ptile]
        ps0 = pq * spp_t[ptile]
        pnsp = torch.minimum(spp_t[ptile], nstrip_t[ptile] - ps0)
        pk0 = first[ptile] + ps0 * w_t[ptile]
        pk1 = torch.minimum(pk0 + pnsp * w_t[ptile], first[ptile] + usz_s[ptile])
        psb = sbase_t[ptile] + ps0
        cut_t = np_t > 1
        part_t = torch.where(cut_t, np_t * R, torch.zeros_like(np_t))
        tpart = torch.cumsum(part_t, 0) - part_t
        pout = torch.where(cut_t[ptile], tpart[ptile] + pq * R, torch.full_like(pq, -1))
        self.npartial = int(part_t.sum().item())
        # entry blocks: (piece, wave, strip), contiguous per (piece, wave); inside a block by row, then column; a block is
        # a whole number of lines (at least one: an empty block still tells the wave to move on to the next strip)
        nblocks = nstrips * NW
        if moved:
            e_p = pbase_t.int()[e_tile] + e_strip // spp_t.clamp(min=1).int()[e_tile]
            bidx = psb.int()[e_p] * NW + e_wave * pnsp.int()[e_p] + (e_strip - ps0.int()[e_p])
            del e_p, e_strip, e_wave, e_tile
            # rank order = (row, column) order inside every block already: a stable sort by block number finishes it
            sb, perm = torch.sort(bidx, stable=True)
            bcnt = torch.bincount(bidx, minlength=nblocks)
            bfirst = torch.cumsum(bcnt, 0) - bcnt
            del bidx
            rank_in = torch.arange(moved, **i32) - bfirst.int()[sb]           # position inside the block
            del bfirst
        else:
            bcnt = torch.zeros(nblocks, **i64)
        FIRST = LINE - 4                                           # a block's first line: 28 entries + 4 column ids
        nl = 1 + torch.clamp(bcnt - FIRST + LINE - 1, min=0) // LINE
        lstart = torch.cumsum(nl, 0) - nl
        nline = int(nl.sum().item()) if nblocks else 0
        # a line = 64 dwords: the 32 entries' meta words, then their 32 values
        ent = torch.empty((nline + 8), 2, LINE, dtype=torch.int32, device=dev)  # (+ slack: the kernel requests up to six lines ahead)
        ent[:, 0, :] = WMAX << 11                                 # pad entries: the all-zero slice ...
        ent[:, 1, :] = -2147483648                                # ... times -0.0: fma(-0.0, +0.0, acc) == acc for every acc
        flat = ent.view(-1)
        if moved:
            if (nline + 8) * 2 * LINE >= 2 ** 31:
                raise RuntimeError('StripPlan: entry stream exceeds int32 indexing')
            later = torch.clamp(rank_in - FIRST, min=0)
            l32 = lstart.int()[sb]
            del sb
            pos = torch.where(rank_in < FIRST, l32 * (2 * LINE) + rank_in, (l32 + 1 + later // LINE) * (2 * LINE) + later % LINE)
            del l32, later, rank_in
            flat[pos] = (e_slot[perm] << 11) | (e_rsub[perm] * 4)
            del e_slot, e_rsub
            flat[pos + LINE] = e_val[perm].view(torch.int32)
            del perm, pos, e_val
        if nline:
            lblk = torch.repeat_interleave(torch.arange(nblocks, **i64), nl)
            l_in = torch.arange(nline, **i64) - lstart[lblk]
            rem = torch.where(l_in == 0, torch.clamp(bcnt[lblk], max=FIRST),
                              torch.clamp(bcnt[lblk] - FIRST - LINE * (l_in - 1), min=0, max=LINE))
            ng = torch.clamp((rem + 3) // 4, min=1)
            hdr = (ng << 20) | ((l_in == nl[lblk] - 1).long() << 24)
            flat[torch.arange(nline, **i64) * (2 * LINE)] |= hdr.int()
            del lblk, l_in, rem, ng, hdr
            # the first line of the block of (piece, wave w, strip sp) names the columns of slots 4 w .. 4 w + 3 of strip
            # sp + 2, in the meta words of its last four entry slots (a first line holds at most 28 entries)
            bp = torch.repeat_interleave(torch.arange(npiece, **i64), pnsp * NW)      # blocks are numbered piece after piece
            within = torch.arange(nblocks, **i64) - (psb * NW)[bp]
            bw, bsp = within // pnsp[bp], within % pnsp[bp]
            kk = pk0[bp][:, None] + ((bsp + 2) * w_t[ptile][bp] + bw * 4)[:, None] + torch.arange(4, **i64)[None, :]
            kk = torch.minimum(torch.maximum(kk, pk0[bp][:, None]), (pk1[bp] - 1)[:, None])
            ids = self.ucol[kk.view(-1)].view(-1, 4)
            at = (lstart * (2 * LINE) + LINE - 4)[:, None] + torch.arange(4, **i64)[None, :]
            flat[at.view(-1)] = ids.view(-1)
            del bp, within, bw, bsp, kk, ids, at
        self.ent = ent
        self.nline = nline
        self.padding = nline * LINE / max(moved, 1)
        ebase = lstart[(psb * NW)[:, None] + torch.arange(NW, **i64)[None, :] * pnsp[:, None]] if npiece else torch.zeros(0, NW, **i64)
        rows = torch.full((ntile * R,), -1, **i64)
        rows[:n] = order
        self.rows = rows.int().contiguous()
        # pieces in working order: most lines first
        if npiece:
            plines = torch.zeros(npiece, **i64)
            strip_lines = nl.view(-1)                              # per block; sum over a piece's blocks
            pid_of_block = torch.repeat_interleave(torch.arange(npiece, **i64), pnsp * NW)
            # blocks are numbered piece after piece only if pieces are in tile order, which they are (psb ascends with the piece id)
            plines.index_add_(0, pid_of_block, strip_lines)
            o = torch.argsort(plines, descending=True, stable=True)
        else:
            o = torch.zeros(0, **i64)
        zero = torch.zeros_like(pq)
        self.desc = torch.stack([ptile, pk0, pk1, pout, pnsp, w_t[ptile], zero, zero], dim=1)[o].int().contiguous()
        self.ebase = ebase[o].contiguous()
        self.npiece, self.ntile = npiece, ntile
        # cut list: one entry per (cut tile, member row): {row, first partial row, pieces, R - 1}
        ct = torch.nonzero(cut_t).flatten()
        crow = rows.view(ntile, R)[ct]
        cbase = tpart[ct][:, None] + torch.arange(R, **i64)[None, :]
        ccnt = np_t[ct][:, None].expand(-1, R)
        okc = crow >= 0
        self.cut = torch.stack([crow[okc], cbase[okc], ccnt[okc], torch.full_like(crow[okc], R - 1)], dim=1).int().contiguous()
        self.ncut = int(self.cut.size(0))
        in_tile = torch.zeros(ntile * R, dtype=torch.bool, device=dev)
        in_tile.view(ntile, R)[ok_t] = True
        self._strip_rows = torch.zeros(n, dtype=torch.bool, device=dev)
        self._strip_rows[order] = in_tile[:n]
        self._cut_rows = torch.zeros(n, dtype=torch.bool, device=dev)
        self._cut_rows[crow[okc]] = True
        self.tiles = int(ok_t.sum().item())
        self.nbytes = 4 * (self.desc.numel() + self.rows.numel() + self.ucol.numel() + self.ent.numel()
                           + 2 * self.ebase.numel() + self.cut.numel())
        self.c = _CStripPlan(R, WMAX, NW, LINE, ntile, npiece, self.ncut, 0, self.desc.data_ptr(), self.rows.data_ptr(),
                             self.ucol.data_ptr(), self.ent.data_ptr(), self.ebase.data_ptr(), self.cut.data_ptr(), nu, nline,
                             self.npartial)
        self._partial = {}

    partial = SegPlan.partial

    def strip_rows(self):
        """Rows whose entries all run through the strip kernel (the others are whole in ``self.rest``)."""
        return self._strip_rows

    def cut_rows(self):
        """Rows of tiles whose union list is cut into pieces (fixed-order piece sums instead of one chain)."""
        return self._cut_rows


class StripHybridPlan:
    """A block as row strips + the rest: ``rest_plan`` (a plan of ``strip.rest``) runs first, with the call's beta /
    overwrite (it lists every row); ``dgnn_spmm_strip_f32`` then adds the strip tiles. A row is whole in one of the two."""

    def __init__(self, strip, rest_plan, rest_fresh=None):
        """``rest_fresh``: a SegPlan of the rest that skips the tiles' rows, cut like ``rest_plan`` -- with it a product that
        overwrites C writes every row once (dgnn_spmm_strip_overwrite_f32) instead of zero-filling the tiles' rows first."""
        self.strip, self.rest_plan, self.rest_fresh = strip, rest_plan, rest_fresh

    def uncut_rows(self):
        """Rows whose result is one strict chain in ascending column order."""
        return self.rest_plan.uncut_rows() & ~self.strip.cut_rows()

    strict_rows = uncut_rows


def spmm_strip(S, B, C, alpha=1.0, fresh=False):
    """C += (strip tiles of S) @ B[:, :F]; B a padded feature block (pad_rows); alpha must be 1.
    ``fresh``: C[rows of the tiles] = ... instead, written without being read; other rows untouched."""
    bptr, ldb = _dense(B, 'B')
    cptr, ldc = _dense(C, 'C')
    F = C.size(1)
    dev = C.device
    P = S.partial(ldb, dev)
    args = (ctypes.byref(S.c), S.rest.n, S.rest.m, bptr, ldb, cptr, ldc, P.data_ptr(), ldb, F, float(alpha))
    if torch.cuda.current_device() == dev.index:
        rc = (_lib.dgnn_spmm_strip_overwrite_f32 if fresh else _lib.dgnn_spmm_strip_f32)(*args, _stream(C))
    else:
        with torch.cuda.device(dev):
            rc = (_lib.dgnn_spmm_strip_overwrite_f32 if fresh else _lib.dgnn_spmm_strip_f32)(*args, _stream(C))
    _check(rc, 'spmm_strip')


def _strip_for(A):
    """The StripPlan of a block if row strips pay for it, else None (decided once per block). DGNN_STRIP=0 never,
    DGNN_STRIP=1 whenever any tile qualifies, default: blocks with at least 32 nonzeros per row and 4 M nonzeros whose
    selected tiles take at least a third of the nonzeros."""
    want = os.environ.get('DGNN_STRIP', 'auto')
    min_nnz = int(os.environ.get('DGNN_STRIP_MIN_NNZ', 1 << 22))
    min_deg = float(os.environ.get('DGNN_STRIP_MIN_DEG', 32))
    if want == '0' or A.nnz < max(min_nnz, 1) or float(A.nnz) / max(A.n, 1) < min_deg:
        return None
    if '_strip_plan' not in A.__dict__:
        S = StripPlan(A)
        if S.npiece == 0 or (want != '1' and S.share < 1.0 / 3):
            S = None
        A.__dict__['_strip_plan'] = S
    return A.__dict__['_strip_plan']


def _dense_for(A):
    """The DensePlan of a block if dense tiles pay for it, else None (decided once per block). DGNN_DENSE=0 never,
    DGNN_DENSE=1 whenever any tile lists something, default: when the tiles take at least a fifth of the nonzeros."""
    want = os.environ.get('DGNN_DENSE', 'auto')
    # blocks under ~8 M nonzeros (the blocks of a 4- or 8-rank partition) are too small for it: a few hundred pieces do
    # not fill the chip (measured on an 8-way Reddit-shaped partition: 8 block products 1.8 -> 2.8 ms, while the rank's
    # 14 M-nonzero row slab gains 10 %)
    min_nnz = int(os.environ.get('DGNN_DENSE_MIN_NNZ', 1 << 23))       # (tests lower both floors to run small models on tiles)
    min_deg = float(os.environ.get('DGNN_DENSE_MIN_DEG', 32))
    if want == '0' or A.nnz < max(min_nnz, 1) or float(A.nnz) / max(A.n, 1) < min_deg:
        return None
    if '_dense_plan' not in A.__dict__:
        D = DensePlan(A)
        if D.nk == 0 or (want != '1' and D.share < 0.2):
            D = None
        A.__dict__['_dense_plan'] = D
    return A.__dict__['_dense_plan']
