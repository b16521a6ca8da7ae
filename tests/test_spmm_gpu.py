"""GPU parity of the HIP SpMM (through the C ABI) against the CPU oracle. Bit-exact where the
kernel keeps the oracle's ascending accumulation order; 1e-5 relative otherwise."""
import ctypes

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

RTOL = 1e-5   # north_star: fp32 aggregation within 1e-5 relative


def _rand_csr(rng, n, m, nnz, hub_rows=0):
    key = rng.integers(0, n * m, nnz)
    if hub_rows:
        hub = rng.integers(0, n, hub_rows)
        extra = np.concatenate([h * m + rng.choice(m, size=min(m, 3000), replace=False) for h in hub])
        key = np.concatenate([key, extra])
    key = np.unique(key)
    return (key // m).astype(np.int32), (key % m).astype(np.int32), rng.random(key.size, dtype=np.float32) - 0.3


def _check(got, want, rowptr, col, val, B, C0, alpha, beta):
    from oracle import ref_cpu
    if np.array_equal(got, want):
        return True
    exact, abs_sum = ref_cpu.spmm_csr_f64(rowptr, col, val, B, C0, alpha, beta)
    bound = RTOL * abs_sum + 1e-30
    assert (np.abs(got - exact) <= bound).all(), f'max err/bound {(np.abs(got - exact) / bound).max()}'
    assert np.abs(got - want).max() <= RTOL * max(np.abs(want).max(), 1e-30)
    return False


CASES = [
    # n, m, nnz, F, alpha, beta
    (300, 280, 4000, 100, 1.0, 1.0),
    (1000, 900, 60000, 602, 1.0, 1.0),
    (1000, 1000, 30000, 256, 1.0, 0.0),
    (777, 555, 20000, 41, 1.0, 1.0),
    (512, 512, 9000, 47, 0.5, 2.0),
    (64, 4096, 3000, 1433, 1.0, 1.0),
    (2000, 2000, 50000, 128, 1.0, 1.0),
    (333, 333, 5000, 172, 1.0, 1.0),
    (100, 100, 2000, 7, 1.0, 1.0),
    (100, 100, 2000, 16, 1.0, 1.0),
    (100, 100, 2000, 1, 1.0, 1.0),
    (129, 67, 1500, 30, -1.0, 0.25),
    (50, 50, 400, 2050, 1.0, 1.0),
]


@pytest.mark.parametrize('n,m,nnz,F,alpha,beta', CASES)
def test_spmm_csr_matches_oracle(n, m, nnz, F, alpha, beta):
    import spmm_cpp
    from oracle import ref_cpu
    rng = np.random.default_rng(n * 7 + F)
    row, col, val = _rand_csr(rng, n, m, nnz)
    B = rng.standard_normal((m, F), dtype=np.float32)
    C0 = rng.standard_normal((n, F), dtype=np.float32)
    if beta == 0.0:
        C0[:] = np.nan   # beta == 0 must not read C
    want = ref_cpu.spmm_coo(row, col, val, n, m, B, C0.copy(), alpha, beta)
    dev = torch.device('cuda:0')
    Cg = torch.from_numpy(C0).to(dev)
    spmm_cpp.spmm_cusparse(torch.from_numpy(row).to(dev), torch.from_numpy(col).to(dev),
                           torch.from_numpy(val).to(dev), n, m, torch.from_numpy(B).to(dev), Cg, alpha, beta)
    got = Cg.cpu().numpy()
    rowptr = ref_cpu.coo2csr(row, n)
    _check(got, want, rowptr, col, val, B, np.nan_to_num(C0), alpha, beta)


def test_hub_rows_and_empty_rows():
    import spmm_cpp
    from oracle import ref_cpu
    rng = np.random.default_rng(5)
    n = m = 6000
    row, col, val = _rand_csr(rng, n, m, 30000, hub_rows=3)
    keep = (row < 100) | (row >= 400)            # rows 100..399 empty
    row, col, val = row[keep], col[keep], val[keep]
    for F in (64, 602):
        B = rng.standard_normal((m, F), dtype=np.float32)
        C0 = rng.standard_normal((n, F), dtype=np.float32)
        want = ref_cpu.spmm_coo(row, col, val, n, m, B, C0.copy())
        dev = torch.device('cuda:0')
        Cg = torch.from_numpy(C0).to(dev)
        spmm_cpp.spmm_cusparse(torch.from_numpy(row).to(dev), torch.from_numpy(col).to(dev),
                               torch.from_numpy(val).to(dev), n, m, torch.from_numpy(B).to(dev), Cg, 1.0, 1.0)
        _check(Cg.cpu().numpy(), want, ref_cpu.coo2csr(row, n), col, val, B, C0, 1.0, 1.0)


def test_empty_matrix_and_degenerate_shapes():
    import spmm_cpp
    dev = torch.device('cuda:0')
    e32 = torch.empty(0, dtype=torch.int32, device=dev)
    ef = torch.empty(0, dtype=torch.float32, device=dev)
    B = torch.rand(5, 8, device=dev)
    C = torch.rand(4, 8, device=dev)
    keep = C.clone()
    spmm_cpp.spmm_cusparse(e32, e32, ef, 4, 5, B, C, 1.0, 1.0)
    assert torch.equal(C, keep)
    spmm_cpp.spmm_cusparse(e32, e32, ef, 4, 5, B, C, 1.0, 0.5)
    assert torch.equal(C, keep * 0.5)
    spmm_cpp.spmm_cusparse(e32, e32, ef, 4, 5, B, C, 1.0, 0.0)
    assert torch.equal(C, torch.zeros_like(C))


def test_strided_operands_and_csr_block():
    import spmm_cpp
    from oracle import ref_cpu
    rng = np.random.default_rng(9)
    n, m, F = 400, 300, 100
    row, col, val = _rand_csr(rng, n, m, 8000)
    B = rng.standard_normal((m, F), dtype=np.float32)
    C0 = rng.standard_normal((n, F), dtype=np.float32)
    want = ref_cpu.spmm_coo(row, col, val, n, m, B, C0.copy())
    dev = torch.device('cuda:0')
    A = torch.sparse_coo_tensor(torch.from_numpy(np.stack([row, col]).astype(np.int64)), torch.from_numpy(val),
                                (n, m)).coalesce().to(dev)
    blk = spmm_cpp.CsrBlock.from_coo(A)
    assert np.array_equal(blk.rowptr.cpu().numpy(), ref_cpu.coo2csr(row, n))      # integer work: bit-exact
    assert np.array_equal(blk.col.cpu().numpy(), col)
    Bpad = torch.zeros(m, F + 28, device=dev); Bpad[:, :F] = torch.from_numpy(B).to(dev)
    Cpad = torch.full((n, F + 4), 7.0, device=dev); Cpad[:, :F] = torch.from_numpy(C0).to(dev)
    spmm_cpp.spmm_csr(blk, Bpad[:, :F], Cpad[:, :F], 1.0, 1.0)
    assert np.array_equal(Cpad[:, :F].cpu().numpy(), want) or np.allclose(Cpad[:, :F].cpu().numpy(), want, rtol=RTOL, atol=1e-6)
    assert (Cpad[:, F:] == 7.0).all()     # padding columns untouched


def test_errors_are_loud():
    import spmm_cpp
    dev = torch.device('cuda:0')
    r = torch.zeros(3, dtype=torch.int32, device=dev)
    v = torch.ones(3, device=dev)
    B = torch.rand(4, 8, device=dev)
    C = torch.rand(4, 8, device=dev)
    with pytest.raises(RuntimeError):
        spmm_cpp.spmm_cusparse(r.long(), r, v, 4, 4, B, C, 1, 1)          # int64 indices
    with pytest.raises(RuntimeError):
        spmm_cpp.spmm_cusparse(r, r, v, 4, 4, B.cpu(), C, 1, 1)           # CPU operand: no CPU path
    with pytest.raises(RuntimeError):
        spmm_cpp.spmm_cusparse(r, r, v, 4, 4, B, B, 1, 1)                 # B aliases C
    with pytest.raises(RuntimeError):
        spmm_cpp.spmm_cusparse(r, r, v, 4, 5, B, C, 1, 1)                 # shape mismatch
    rc = spmm_cpp.lib.dgnn_spmm_csr_f32(None, None, None, 4, 4, 3, None, 8, None, 8, 8,
                                        ctypes.c_float(1), ctypes.c_float(1), None)
    assert rc != 0 and spmm_cpp.lib.dgnn_last_error()
