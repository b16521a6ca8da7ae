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
import os

import torch

_LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'lib', 'libdgnn_spmm.so')
if not os.path.exists(_LIB_PATH):
    raise RuntimeError(
        f'spmm_cpp: {os.path.normpath(_LIB_PATH)} is missing -- build it with '
        '`python -c "import __graft_entry__ as g; g.build()"` (hipcc --offload-arch=gfx950). '
        'There is no CPU fallback.')
_lib = ctypes.CDLL(_LIB_PATH)

_p, _i32, _i64, _f32 = ctypes.c_void_p, ctypes.c_int32, ctypes.c_int64, ctypes.c_float
_lib.dgnn_version.restype = ctypes.c_int
_lib.dgnn_last_error.restype = ctypes.c_char_p
_lib.dgnn_coo2csr_i32.argtypes = [_p, _i64, _i32, _p, _p]
_lib.dgnn_idx_i64_to_i32.argtypes = [_p, _i64, _p, _p]
_lib.dgnn_spmm_csr_f32.argtypes = [_p, _p, _p, _i32, _i32, _i64, _p, _i64, _p, _i64, _i32, _f32, _f32, _p]
_lib.dgnn_spmm_coo_f32.argtypes = [_p, _p, _p, _i32, _i32, _i64, _p, _p, _i64, _p, _i64, _i32, _f32, _f32, _p]
_lib.dgnn_set_variant.argtypes = [ctypes.c_int]
for _f in ('dgnn_coo2csr_i32', 'dgnn_idx_i64_to_i32', 'dgnn_spmm_csr_f32', 'dgnn_spmm_coo_f32',
           'dgnn_set_variant', 'dgnn_get_variant'):
    getattr(_lib, _f).restype = ctypes.c_int

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
    if B.size(0) != m or C.size(0) != n or B.size(1) != F:
        raise RuntimeError(f'spmm_cpp: shape mismatch A[{n}x{m}] B{tuple(B.shape)} C{tuple(C.shape)}')
    if not (A_values.device == A_row_idx.device == A_col_idx.device == B.device == C.device):
        raise RuntimeError('spmm_cpp: operands are on different devices')
    with torch.cuda.device(C.device):
        ws = torch.empty(n + 1, dtype=torch.int32, device=C.device)
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
        return cls(n, m, rowptr, col32, val.contiguous())


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


def set_variant(v):
    _check(_lib.dgnn_set_variant(int(v)), 'set_variant')
