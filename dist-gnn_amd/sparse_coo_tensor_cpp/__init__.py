"""sparse_coo_tensor_cpp -- the import name of the older trainers' extension
(1D_CAGNET/sparse-extension/setup.py:4-5; `from sparse_coo_tensor_cpp import sparse_coo_tensor_gpu, spmm_gpu`,
1D_CAGNET/dist_1d.py:11, dist_15d.py:11), over the same C ABI as spmm_cpp.

  spmm_gpu(row_i32, col_i32, val_f32, n, m, B, C)      C += A @ B in place, alpha = beta = 1
                                                        (sparse_coo_tensor.cpp:115-196)
  sparse_coo_tensor_gpu(indices, values, size)          sparse COO tensor built without validating the indices,
                                                        i.e. without the device->host synchronisation of the
                                                        checked constructor (sparse_coo_tensor.cpp:45-56)
"""
import torch

from spmm_cpp import spmm_gpu  # noqa: F401


def sparse_coo_tensor_gpu(indices, values, size):
    if values.dim() == 0:                       # expand_values_if_needed, sparse_coo_tensor.cpp:35-43
        values = values.expand(1)
    return torch.sparse_coo_tensor(indices, values, tuple(size), check_invariants=False)
