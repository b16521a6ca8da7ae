"""dgnn_spmm_coo_noalloc_f32 (the entry point that allocates nothing: its row pointer, work plan and partial rows
live in a file-scope device array) against dgnn_spmm_coo_aligned_f32 with caller workspaces, through the C ABI:
bit-identical outputs on two calls in a row, the second on a smaller input (the array keeps the first call's
contents); sizes beyond the array are rejected. Collected last: a failure here hides no other test."""
import ctypes

import pytest
import torch

pytestmark = pytest.mark.gpu


def _block(n, m, nnz, F, seed, dev, hub=0):
    g = torch.Generator().manual_seed(seed)
    row = torch.randint(0, n, (nnz,), generator=g)
    if hub:                                               # a few very long rows: cut into pieces -> partial rows + merge
        row[:hub] = torch.randint(0, 3, (hub,), generator=g)
    col = torch.randint(0, m, (nnz,), generator=g)
    A = torch.sparse_coo_tensor(torch.stack([row, col]), torch.rand(nnz, generator=g), (n, m)).coalesce()
    idx = A.indices()
    ld = (F + 31) // 32 * 32
    B = torch.zeros(m, ld)
    B[:, :F] = torch.rand(m, F, generator=g)
    C = torch.rand(n, F, generator=g)
    return (idx[0].int().to(dev), idx[1].int().to(dev), A.values().to(dev), B.to(dev), C.to(dev), ld)


def _aligned(lib, spmm_cpp, row, col, val, n, m, B, ld, C, F, alpha, beta):
    nnz = val.numel()
    ws_ints, prow = ctypes.c_int64(), ctypes.c_int64()
    assert lib.dgnn_coo_aligned_workspace(n, m, nnz, ld, ctypes.byref(ws_ints), ctypes.byref(prow)) == 0
    rp = torch.empty(n + 1, dtype=torch.int32, device=C.device)
    ws = torch.empty(max(ws_ints.value, 4), dtype=torch.int32, device=C.device)
    part = torch.empty(max(prow.value, 1) * ld, dtype=torch.float32, device=C.device)
    rc = lib.dgnn_spmm_coo_aligned_f32(row.data_ptr(), col.data_ptr(), val.data_ptr(), n, m, nnz, rp.data_ptr(), ws.data_ptr(),
                                       B.data_ptr(), ld, C.data_ptr(), C.stride(0), part.data_ptr(), ld, F, alpha, beta, None)
    assert rc == 0, lib.dgnn_last_error()


def _noalloc(lib, row, col, val, n, m, B, ld, C, F, alpha, beta):
    return lib.dgnn_spmm_coo_noalloc_f32(row.data_ptr(), col.data_ptr(), val.data_ptr(), n, m, val.numel(),
                                         B.data_ptr(), ld, C.data_ptr(), C.stride(0), F, alpha, beta, None)


def test_noalloc_entry_point_equals_the_workspace_one_twice_in_a_row():
    import spmm_cpp
    lib = spmm_cpp.lib
    dev = torch.device('cuda:0')
    torch.cuda.set_device(dev)
    # first a large call (long rows -> pieces, partial rows, merges; 3 column-block stages), then a smaller one that
    # finds the first call's lists and partial rows still in the array
    for n, m, nnz, F, hub, beta in ((60000, 250000, 3_000_000, 100, 40000, 1.0), (700, 900, 20000, 41, 6000, 0.5),
                                    (50, 64, 0, 8, 0, 0.0)):
        row, col, val, B, C0, ld = _block(n, m, nnz, F, n, dev, hub)
        need = ctypes.c_int64()
        assert lib.dgnn_noalloc_scratch_needed(n, m, val.numel(), ld, ctypes.byref(need)) == 0
        want, got = C0.clone(), C0.clone()
        _aligned(lib, spmm_cpp, row, col, val, n, m, B, ld, want, F, 1.5, beta)
        before = torch.cuda.memory_allocated(dev)
        assert _noalloc(lib, row, col, val, n, m, B, ld, got, F, 1.5, beta) == 0, lib.dgnn_last_error()
        torch.cuda.synchronize()
        assert torch.cuda.memory_allocated(dev) == before
        assert torch.equal(got, want), (n, m, nnz, F, (got - want).abs().max().item())
        assert torch.isfinite(got).all()


def test_noalloc_entry_point_rejects_what_does_not_fit():
    import spmm_cpp
    lib = spmm_cpp.lib
    dev = torch.device('cuda:0')
    cap = ctypes.c_int64()
    assert lib.dgnn_noalloc_limits(ctypes.byref(cap)) == 0 and cap.value == 64 << 20
    n = m = 5_000_000                                     # row pointer and plan lists alone exceed the array
    need = ctypes.c_int64()
    assert lib.dgnn_noalloc_scratch_needed(n, m, 10, 32, ctypes.byref(need)) == 0 and need.value > cap.value
    z = torch.zeros(16, dtype=torch.int32, device=dev)
    f = torch.zeros(64, device=dev)
    rc = lib.dgnn_spmm_coo_noalloc_f32(z.data_ptr(), z.data_ptr(), f.data_ptr(), n, m, 10, f.data_ptr(), 32, f.data_ptr(), 32, 8,
                                       1.0, 1.0, None)
    assert rc == 1 and b'scratch' in lib.dgnn_last_error()           # DGNN_E_BADARG before anything is launched
    rc = lib.dgnn_spmm_coo_noalloc_f32(z.data_ptr(), z.data_ptr(), f.data_ptr(), 4, 4, 2, f.data_ptr() + 4, 30, f.data_ptr(), 32, 8,
                                       1.0, 1.0, None)
    assert rc == 1                                                     # misaligned / unpadded B


def test_noalloc_entry_point_under_graph_capture():
    """What the entry point is for: after the per-device set-up (dgnn_noalloc_limits, outside the capture) a call only
    enqueues kernels on the stream it is given -- captured into a HIP graph and replayed on new operand VALUES (same
    buffers), it gives the bits of the eager workspace entry point each time."""
    import spmm_cpp
    lib = spmm_cpp.lib
    dev = torch.device('cuda:0')
    torch.cuda.set_device(dev)
    cap = ctypes.c_int64()
    assert lib.dgnn_noalloc_limits(ctypes.byref(cap)) == 0           # set-up of this device, outside the capture
    n, m, nnz, F, hub = 5000, 7000, 200000, 100, 30000               # long rows: pieces, partial rows, merge
    row, col, val, B, C0, ld = _block(n, m, nnz, F, 77, dev, hub)
    out = C0.clone()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        rc = lib.dgnn_spmm_coo_noalloc_f32(row.data_ptr(), col.data_ptr(), val.data_ptr(), n, m, val.numel(), B.data_ptr(), ld,
                                           out.data_ptr(), out.stride(0), F, 1.0, 0.5, torch.cuda.current_stream().cuda_stream)
    assert rc == 0, lib.dgnn_last_error()
    for scale in (1.0, 3.0):
        B.mul_(scale)
        out.copy_(C0)
        want = C0.clone()
        _aligned(lib, spmm_cpp, row, col, val, n, m, B, ld, want, F, 1.0, 0.5)
        graph.replay()
        torch.cuda.synchronize()
        assert torch.equal(out, want), (scale, (out - want).abs().max().item())
