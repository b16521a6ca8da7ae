"""GPU parity of the HIP SpMM (through the C ABI) against the CPU oracle. Bit-exact where the
kernel keeps the oracle's ascending accumulation order; 1e-5 relative otherwise."""
import ctypes
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

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


# --------------------------------------------------------------------------- fast path (work plan)
def _gpu_block(row, col, val, n, m):
    import spmm_cpp
    dev = torch.device('cuda:0')
    A = torch.sparse_coo_tensor(torch.from_numpy(np.stack([row, col]).astype(np.int64)), torch.from_numpy(val),
                                (n, m)).coalesce().to(dev)
    return spmm_cpp.CsrBlock.from_coo(A)


PLAN_CASES = [
    # n, m, nnz, hubs, F, nstage, piece_len, alpha, beta
    (1000, 900, 60000, 0, 602, 1, 128, 1.0, 1.0),
    (1000, 900, 60000, 2, 602, 3, 256, 1.0, 1.0),
    (1000, 900, 60000, 2, 256, 3, 128, 1.0, 1.0),
    (1500, 1500, 40000, 1, 100, 2, 100, 1.0, 1.0),
    (3000, 3000, 90000, 3, 256, 4, 64, 1.0, 1.0),
    (777, 555, 20000, 0, 41, 2, 256, 1.0, 1.0),
    (512, 512, 9000, 1, 47, 1, 32, 0.5, 2.0),
    (300, 5000, 30000, 0, 100, 5, 256, 1.0, 0.0),
    (333, 333, 5000, 0, 172, 2, 256, 1.0, 1.0),
    (100, 100, 2000, 0, 7, 1, 256, 1.0, 1.0),
    (100, 100, 2000, 0, 16, 2, 16, -1.0, 0.25),
    (64, 4096, 3000, 0, 1433, 4, 256, 1.0, 1.0),
    (4000, 100, 50000, 0, 64, 1, 8, 1.0, 1.0),
    # runs longer than one LDS strip (256 entries for 16-lane groups, 128 for 8-lane groups): walked strip by strip
    (1000, 900, 60000, 2, 602, 1, 4096, 1.0, 1.0),
    (3000, 3000, 90000, 3, 256, 1, 1000000, 1.0, 1.0),      # nothing cut: every row must carry the oracle's bits
    (2000, 6000, 120000, 3, 41, 2, 1024, 1.0, 1.0),
    (1500, 5000, 100000, 2, 100, 1, 300, 0.5, 1.0),         # piece length that is no multiple of the strip
    (900, 7000, 80000, 2, 1433, 3, 700, 1.0, 0.0),
]


@pytest.mark.parametrize('n,m,nnz,hubs,F,nstage,piece,alpha,beta', PLAN_CASES)
def test_plan_path_matches_oracle(n, m, nnz, hubs, F, nstage, piece, alpha, beta):
    import spmm_cpp
    from oracle import ref_cpu
    rng = np.random.default_rng(n + 13 * F + nstage)
    row, col, val = _rand_csr(rng, n, m, nnz, hub_rows=hubs)
    B = rng.standard_normal((m, F), dtype=np.float32)
    C0 = rng.standard_normal((n, F), dtype=np.float32)
    if beta == 0.0:
        C0[:] = np.nan
    want = ref_cpu.spmm_coo(row, col, val, n, m, B, C0.copy(), alpha, beta)
    dev = torch.device('cuda:0')
    blk = _gpu_block(row, col, val, n, m)
    plan = spmm_cpp.SegPlan(blk, nstage, piece)
    Bp = spmm_cpp.pad_rows(torch.from_numpy(B).to(dev))
    assert Bp.shape == (m, spmm_cpp.padded_width(F)) and (Bp[:, F:] == 0).all()
    Cg = torch.from_numpy(C0).to(dev)
    spmm_cpp.spmm_plan(blk, Bp, Cg, alpha, beta, plan)
    got = Cg.cpu().numpy()
    rowptr = ref_cpu.coo2csr(row, n)
    exact = _check(got, want, rowptr, col, val, B, np.nan_to_num(C0), alpha, beta)
    if plan.ncut == 0:
        assert exact, 'uncut plans keep the oracle accumulation chain: must be bit-identical'
    else:   # only rows that were cut may differ from the oracle's bits
        cut_rows = np.unique(plan.cut.cpu().numpy()[:, 0])
        mask = np.ones(n, dtype=bool); mask[cut_rows] = False
        assert np.array_equal(got[mask], want[mask])


@pytest.mark.parametrize('builder', ['torch', 'device'])
@pytest.mark.parametrize('n,m,nnz,hubs,F,nstage,piece', [c[:7] for c in PLAN_CASES[:6] + PLAN_CASES[13:16]])
def test_overwriting_product_needs_no_initialised_output(n, m, nnz, hubs, F, nstage, piece, builder):
    """dgnn_spmm_plan_overwrite_f32 on a C full of NaN (contiguous, and as a view of a wider buffer) against
    dgnn_spmm_plan_f32(beta = 1) on a zero-filled C: every row is stored by the first stage (empty rows and rows
    with nothing in the first column block included), nothing is read, nothing past column F is touched."""
    import spmm_cpp
    rng = np.random.default_rng(5 * n + F)
    row, col, val = _rand_csr(rng, n, m, nnz, hub_rows=hubs)
    row[row == 3] = 4                                 # an empty row, and rows with nothing in the first column block
    order = np.lexsort((col, row)); row, col, val = row[order], col[order], val[order]
    dev = torch.device('cuda:0')
    blk = _gpu_block(row, col, val, n, m)
    plan = spmm_cpp.SegPlan(blk, nstage, piece) if builder == 'torch' else spmm_cpp.DevicePlan(blk, nstage, piece, F)
    assert plan.c.covers_rows == 1
    Bp = spmm_cpp.pad_rows(torch.from_numpy(rng.standard_normal((m, F), dtype=np.float32)).to(dev))
    want = torch.zeros(n, F, device=dev)
    spmm_cpp.spmm_plan(blk, Bp, want, 1.0, 1.0, plan)
    got = torch.full((n, F), float('nan'), device=dev)
    spmm_cpp.spmm_plan(blk, Bp, got, 1.0, 1.0, plan, overwrite=True)
    assert torch.equal(got, want)
    got.fill_(float('nan'))
    spmm_cpp.spmm_plan(blk, Bp, got, 1.0, 0.0, plan)              # beta = 0 takes the same route
    assert torch.equal(got, want)
    wide = torch.full((n, spmm_cpp.padded_width(F) + 32), float('nan'), device=dev)
    spmm_cpp.spmm_plan(blk, Bp, wide[:, :F], 1.0, 1.0, plan, overwrite=True)
    assert torch.equal(wide[:, :F], want) and torch.isnan(wide[:, F:]).all()
    H = Bp[:, :F]
    z = torch.full((n, F), float('nan'), device=dev)
    spmm_cpp.aggregate(blk, H, z, overwrite=True)
    z2 = torch.zeros(n, F, device=dev)
    spmm_cpp.aggregate(blk, H, z2)
    assert torch.equal(z, z2)
    spmm_cpp.aggregate(blk, H, z)                     # second stage: accumulates
    spmm_cpp.aggregate(blk, H, z2, overwrite=False)
    assert torch.equal(z, z2) and not torch.equal(z, want)


def test_older_extension_name_and_signatures():
    """`from sparse_coo_tensor_cpp import sparse_coo_tensor_gpu, spmm_gpu` (1D_CAGNET/dist_1d.py:11): alpha = beta = 1
    product in place, and the unchecked sparse constructor."""
    from sparse_coo_tensor_cpp import sparse_coo_tensor_gpu, spmm_gpu
    from oracle import ref_cpu
    rng = np.random.default_rng(31)
    n, m, F = 400, 350, 48
    row, col, val = _rand_csr(rng, n, m, 9000)
    B = rng.standard_normal((m, F), dtype=np.float32)
    C0 = rng.standard_normal((n, F), dtype=np.float32)
    want = ref_cpu.spmm_coo(row, col, val, n, m, B, C0.copy(), 1.0, 1.0)
    dev = torch.device('cuda:0')
    idx = torch.from_numpy(np.stack([row, col]).astype(np.int64)).to(dev)
    A = sparse_coo_tensor_gpu(idx, torch.from_numpy(val).to(dev), (n, m))
    assert A.is_sparse and tuple(A.shape) == (n, m) and torch.equal(A._indices(), idx)
    C = torch.from_numpy(C0).to(dev)
    spmm_gpu(A._indices()[0].int(), A._indices()[1].int(), A._values(), n, m, torch.from_numpy(B).to(dev), C)
    assert np.array_equal(C.cpu().numpy(), want)
    assert sparse_coo_tensor_gpu(idx[:, :1], torch.tensor(2.0, device=dev), (n, m))._values().shape == (1,)


@pytest.mark.parametrize('kind', ['runs', 'bundles', 'tiles', 'strips'])
def test_plan_spmm_can_be_captured_in_a_hip_graph(kind):
    """The fast-path entry points only launch kernels on the given stream (no allocation, no synchronisation),
    so a caller may capture them in a HIP graph and replay it on new operand contents: run plans, bundle plans, and
    dense tiles + remainder (two entry points back to back)."""
    import spmm_cpp
    rng = np.random.default_rng(77)
    n, m, F = 3000, 2500, 256
    row, col, val = _rand_csr(rng, n, m, 90000, hub_rows=3)
    fold = rng.random(col.size) < 0.5                    # popular columns: something for bundles and tiles to share
    key = np.unique(row.astype(np.int64) * m + np.where(fold, col % 120, col))
    row, col = (key // m).astype(np.int32), (key % m).astype(np.int32)
    val = rng.random(key.size, dtype=np.float32)
    dev = torch.device('cuda:0')
    blk = _gpu_block(row, col, val, n, m)
    if kind == 'runs':
        plan = spmm_cpp.SegPlan(blk, 3, 64)             # three stages, cut rows: stage kernels + merge kernels
    elif kind == 'bundles':
        plan = spmm_cpp.BundlePlan(blk, 3, 4, 64)
    elif kind == 'strips':                               # row strips (cut tiles: partial rows + merge) after the plan of the rest
        S = spmm_cpp.StripPlan(blk, 0.6, 4, min_tile_nnz=64)
        assert S.npiece > 0 and S.ncut > 0                 # (every tile is selected here: the plan of the rest only overwrites C)
        plan = spmm_cpp.StripHybridPlan(S, spmm_cpp.SegPlan(S.rest, 2, 64))
    else:
        D = spmm_cpp.DensePlan(blk, 3, 8, 64)
        assert D.nk > 0 and D.ncut > 0
        plan = spmm_cpp.HybridPlan(D, spmm_cpp.SegPlan(D.rest, 2, 64))
    assert (plan.strip if kind == 'strips' else plan.rest_plan if kind == 'tiles' else plan).ncut > 0
    B = spmm_cpp.pad_rows(torch.rand(m, F, device=dev))
    C = torch.zeros(n, F, device=dev)
    spmm_cpp.spmm_plan(blk, B, C, 1.0, 0.0, plan)       # warm-up: sizes the partial buffer outside the capture
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        spmm_cpp.spmm_plan(blk, B, C, 1.0, 0.0, plan)
    for seed in (1, 2):
        B2 = spmm_cpp.pad_rows(torch.rand(m, F, device=dev, generator=torch.Generator(device=dev).manual_seed(seed)))
        want = torch.full((n, F), float('nan'), device=dev)
        spmm_cpp.spmm_plan(blk, B2, want, 1.0, 0.0, plan)
        B.copy_(B2)
        C.fill_(float('nan'))
        graph.replay()
        torch.cuda.synchronize()
        assert torch.equal(C, want)


def test_plan_is_deterministic_and_covers_every_nonzero():
    import spmm_cpp
    rng = np.random.default_rng(3)
    n = m = 5000
    row, col, val = _rand_csr(rng, n, m, 80000, hub_rows=4)
    blk = _gpu_block(row, col, val, n, m)
    p1, p2 = spmm_cpp.SegPlan(blk, 6, 128), spmm_cpp.SegPlan(blk, 6, 128)
    assert torch.equal(p1.seg, p2.seg) and torch.equal(p1.cut, p2.cut)
    seg = p1.seg.cpu().numpy().astype(np.int64)
    covered = np.zeros(val.size, dtype=np.int32)
    for r, k0, k1, _ in seg:
        covered[k0:k1] += 1
        assert (row[k0:k1] == r).all()
    assert (covered == 1).all()                     # every nonzero in exactly one segment
    assert (seg[:, 2] - seg[:, 1]).max() <= 128
    off = p1.stage_seg_off.numpy()
    cbw = -(-m // 6)
    for s in range(6):                              # stage s only touches column block s, longest first
        part = seg[off[s]:off[s + 1]]
        lens = part[:, 2] - part[:, 1]
        assert (np.diff(lens) <= 0).all()
        for r, k0, k1, _ in part[:50]:
            assert (col[k0:k1] // cbw == s).all()
    Cs = []
    dev = torch.device('cuda:0')
    Bp = spmm_cpp.pad_rows(torch.rand(m, 256, device=dev))
    for _ in range(3):                              # run-to-run identical bits (no atomics)
        C = torch.zeros(n, 256, device=dev)
        spmm_cpp.spmm_plan(blk, Bp, C, 1.0, 0.0, p1)
        Cs.append(C)
    assert torch.equal(Cs[0], Cs[1]) and torch.equal(Cs[0], Cs[2])


def test_aggregate_helper_and_plan_errors():
    import spmm_cpp
    from oracle import ref_cpu
    rng = np.random.default_rng(21)
    n, m, F = 500, 400, 100
    row, col, val = _rand_csr(rng, n, m, 9000)
    blk = _gpu_block(row, col, val, n, m)
    dev = torch.device('cuda:0')
    H = rng.standard_normal((m, F), dtype=np.float32)
    Z0 = rng.standard_normal((n, F), dtype=np.float32)
    Z = torch.from_numpy(Z0).to(dev)
    spmm_cpp.aggregate(blk, torch.from_numpy(H).to(dev), Z)
    assert np.array_equal(Z.cpu().numpy(), ref_cpu.spmm_coo(row, col, val, n, m, H, Z0.copy()))
    with pytest.raises(RuntimeError):               # unpadded, misaligned B is rejected by the fast path
        spmm_cpp.spmm_plan(blk, torch.rand(m, F + 1, device=dev)[:, 1:], Z)
    with pytest.raises(RuntimeError):
        spmm_cpp.spmm_plan(blk, torch.rand(m + 1, 128, device=dev), Z)


@pytest.mark.parametrize('F,ldd', [(1, 4), (41, 44), (41, 64), (100, 128), (602, 608), (256, 256), (7, 32)])
def test_gather_rows_is_an_exact_copy(F, ldd):
    """dgnn_gather_rows_f32: packed rows (repeats and any order allowed), zero fill up to the pitch."""
    import spmm_cpp
    dev = torch.device('cuda:0')
    g = torch.Generator().manual_seed(F * 7 + ldd)
    n, k = 5000, 3777
    src = torch.rand(n, F, generator=g).to(dev)
    idx = torch.randint(0, n, (k,), generator=g).to(torch.int32).to(dev)
    dst = torch.full((k, ldd), float('nan'), device=dev)
    out = spmm_cpp.gather_rows(src, idx, dst)
    assert out.data_ptr() == dst.data_ptr()
    assert torch.equal(dst[:, :F], src[idx.long()])
    assert torch.count_nonzero(dst[:, F:]) == 0 and not torch.isnan(dst).any()
    # strided source (a padded block) and a freshly allocated destination
    wide = torch.rand(n, F + 5, generator=g).to(dev)
    got = spmm_cpp.gather_rows(wide[:, :F], idx)
    assert got.shape == (k, spmm_cpp.padded_width(F)) and torch.equal(got[:, :F], wide[:, :F][idx.long()])
    # nothing to gather: no launch, no error
    spmm_cpp.gather_rows(src, idx[:0], dst[:0])
    with pytest.raises(RuntimeError):
        spmm_cpp.gather_rows(src.cpu(), idx, dst)
    with pytest.raises(RuntimeError):
        spmm_cpp.gather_rows(src, idx.long(), dst)
    with pytest.raises(RuntimeError, match='overlap'):
        spmm_cpp.gather_rows(dst, idx, dst)


_VARIANT_CHILD = r"""
import os, sys
import numpy as np, torch
root = sys.argv[1]
sys.path.insert(0, os.path.join(root, 'dist-gnn_amd')); sys.path.insert(0, os.path.join(root, 'tests'))
import spmm_cpp
from test_spmm_gpu import _rand_csr, _gpu_block
rng = np.random.default_rng(77)
n, m, F = 1500, 2500, 300
row, col, val = _rand_csr(rng, n, m, 70000, hub_rows=3)
blk = _gpu_block(row, col, val, n, m)
dev = torch.device('cuda:0')
Bp = spmm_cpp.pad_rows(torch.from_numpy(rng.standard_normal((m, F), dtype=np.float32)).to(dev))
C0 = torch.from_numpy(rng.standard_normal((n, F), dtype=np.float32)).to(dev)
plan = spmm_cpp.SegPlan(blk, 3, 600)
base = C0.clone()
spmm_cpp.spmm_plan(blk, Bp, base, 1.0, 1.0, plan)
for variant in (2, 3, 8, 16, 32, 102, 106, 108):
    spmm_cpp.set_variant(variant)
    got = C0.clone()
    spmm_cpp.spmm_plan(blk, Bp, got, 1.0, 1.0, plan)
    spmm_cpp.set_variant(0)
    assert torch.equal(got, base), variant
print('VARIANTS AGREE')
"""


def test_kernel_variants_agree_bit_for_bit():
    """The A/B knobs of the benchmark tools (64-bit offsets, contiguous XCD shares, 8/16/32-lane groups, 2/6/8 gathers in
    flight) only change how the work is laid out: same accumulation chain, so the same bits as the default dispatch. The
    knobs exist only in the experiments build of the library (lib/libdgnn_spmm_exp.so, -DDGNN_EXPERIMENTS): a child process
    loads that one through DGNN_LIB; the product library this suite runs on has no such switch."""
    import subprocess
    import sys
    import spmm_cpp
    assert not hasattr(spmm_cpp.lib, 'dgnn_set_variant')                  # the product build carries no knob
    with pytest.raises(RuntimeError, match='experiments build'):
        spmm_cpp.set_variant(2)
    exp = os.path.join(ROOT, 'dist-gnn_amd', 'lib', 'libdgnn_spmm_exp.so')
    assert os.path.exists(exp), 'build it: python -c "import __graft_entry__ as g; g.build()"'
    out = subprocess.run([sys.executable, '-c', _VARIANT_CHILD, ROOT], env=dict(os.environ, DGNN_LIB=exp),
                         capture_output=True, text=True, timeout=300)
    assert 'VARIANTS AGREE' in out.stdout, out.stdout + out.stderr


def test_feature_blocks_beyond_4_gib_use_wide_offsets():
    """B larger than 4 GiB (papers-shaped blocks): the stage kernel switches to 64-bit row offsets on its own.
    Checked against the general row kernel (64-bit addressing throughout) on rows the plan does not cut."""
    import spmm_cpp
    dev = torch.device('cuda:0')
    m, F, n = 1_100_000, 1000, 600                      # padded B: 1.1 M x 1024 floats = 4.5 GB
    g = torch.Generator(device=dev).manual_seed(9)
    B = torch.rand(m, spmm_cpp.padded_width(F), device=dev, generator=g)
    rng = np.random.default_rng(9)
    row = np.repeat(np.arange(n, dtype=np.int32), 40)
    col = np.sort(rng.integers(0, m, (n, 40)), axis=1)
    col[:, -1] = m - 1 - np.arange(n) % 7               # touch the last rows of B: offsets past 2^32 bytes
    col = np.sort(col, axis=1)
    keep = np.concatenate([np.ones((n, 1), bool), col[:, 1:] != col[:, :-1]], axis=1).ravel()
    row, col = row[keep], col.ravel()[keep].astype(np.int32)
    val = rng.random(row.size, dtype=np.float32)
    blk = _gpu_block(row, col, val, n, m)
    want = torch.zeros(n, F, device=dev)
    spmm_cpp.spmm_csr(blk, B[:, :F], want, 1.0, 0.0)
    got = torch.full((n, F), float('nan'), device=dev)
    spmm_cpp.spmm_plan(blk, B, got, 1.0, 0.0, spmm_cpp.SegPlan(blk, 2))
    assert torch.equal(got, want)
    del B
    torch.cuda.empty_cache()


@pytest.mark.parametrize('n,m,nnz,F,alpha,beta', [(2500, 2500, 150000, 602, 1.0, 1.0), (4000, 1500, 90000, 256, 1.0, 0.0),
                                                  (1200, 5000, 120000, 41, 0.5, 2.0), (3000, 3000, 80000, 100, -1.0, 1.0),
                                                  (3000, 140000, 200000, 64, 1.0, 1.0)])     # wide enough for 2 column stages
def test_reference_signature_takes_the_fast_kernel_without_a_plan(n, m, nnz, F, alpha, beta):
    """spmm_cusparse(row, col, val, ...) with enough nonzeros runs the staged kernel on a work plan built on the
    device (dgnn_spmm_coo_aligned_f32; B is padded on the way in). Rows up to the piece length carry the
    oracle's bits; the hub rows are cut and merged (within the fp64 bound), and nothing depends on the run."""
    import spmm_cpp
    from oracle import ref_cpu
    assert nnz >= spmm_cpp._FAST_COO_MIN_NNZ
    rng = np.random.default_rng(n + F)
    row, col, val = _rand_csr(rng, n, m, nnz, hub_rows=2)
    B = rng.standard_normal((m, F), dtype=np.float32)
    C0 = rng.standard_normal((n, F), dtype=np.float32)
    if beta == 0.0:
        C0[:] = np.nan
    want = ref_cpu.spmm_coo(row, col, val, n, m, B, C0.copy(), alpha, beta)
    dev = torch.device('cuda:0')
    Cg = torch.from_numpy(C0).to(dev)
    args = (torch.from_numpy(row).to(dev), torch.from_numpy(col).to(dev), torch.from_numpy(val).to(dev), n, m,
            torch.from_numpy(B).to(dev))
    spmm_cpp.spmm_cusparse(*args, Cg, alpha, beta)
    got = Cg.cpu().numpy()
    rowptr = ref_cpu.coo2csr(row, n)
    _check(got, want, rowptr, col, val, B, np.nan_to_num(C0), alpha, beta)
    uncut = (rowptr[1:] - rowptr[:-1]) <= 128                 # piece length of a block this small
    assert uncut.sum() > n // 2 and np.array_equal(got[uncut], want[uncut])
    again = torch.from_numpy(C0).to(dev)
    spmm_cpp.spmm_cusparse(*args, again, alpha, beta)
    assert torch.equal(again, Cg)                             # list order varies between runs, the result does not


@pytest.mark.parametrize('n,m,nnz,hubs,F,nstage,piece,alpha,beta', [
    (1000, 900, 60000, 2, 602, 3, 256, 1.0, 1.0), (3000, 3000, 90000, 3, 256, 4, 64, 1.0, 1.0),
    (2000, 6000, 120000, 3, 41, 2, 1024, 0.5, 2.0), (900, 7000, 80000, 2, 1433, 16, 700, 1.0, 0.0),
    (1500, 5000, 100000, 2, 100, 1, 1000000, 1.0, 1.0), (64, 4096, 3000, 0, 47, 5, 8, 1.0, 1.0),
    (300, 300, 0, 0, 16, 2, 128, 1.0, 1.0)])
def test_device_built_plan_matches_oracle(n, m, nnz, hubs, F, nstage, piece, alpha, beta):
    """dgnn_plan_build: the plan made on the device (stages by binary search, length classes, cut runs, dead
    slots) drives the same kernels to the same answer: oracle bits on every row no stage cuts, the fp64
    bound on the rest, identical from run to run although the list order is not."""
    import spmm_cpp
    from oracle import ref_cpu
    rng = np.random.default_rng(n + 7 * F + nstage)
    if nnz:
        row, col, val = _rand_csr(rng, n, m, nnz, hub_rows=hubs)
    else:
        row, col, val = np.zeros(0, np.int32), np.zeros(0, np.int32), np.zeros(0, np.float32)
    B = rng.standard_normal((m, F), dtype=np.float32)
    C0 = rng.standard_normal((n, F), dtype=np.float32)
    if beta == 0.0:
        C0[:] = np.nan
    want = ref_cpu.spmm_coo(row, col, val, n, m, B, C0.copy(), alpha, beta)
    dev = torch.device('cuda:0')
    blk = _gpu_block(row, col, val, n, m)
    plan = spmm_cpp.DevicePlan(blk, nstage, piece, F)
    assert plan.nstage == nstage and plan.piece_len == piece
    Bp = spmm_cpp.pad_rows(torch.from_numpy(B).to(dev))
    Cg = torch.from_numpy(C0).to(dev)
    spmm_cpp.spmm_plan(blk, Bp, Cg, alpha, beta, plan)
    got = Cg.cpu().numpy()
    rowptr = ref_cpu.coo2csr(row, n)
    _check(got, want, rowptr, col, val, B, np.nan_to_num(C0), alpha, beta)
    # rows with a run longer than the piece length inside one column block are the only ones allowed to differ
    cbw = -(-m // nstage)
    runs = np.zeros((n, nstage), dtype=np.int64)
    np.add.at(runs, (row, col // cbw), 1)
    uncut = runs.max(axis=1) <= piece
    assert np.array_equal(got[uncut], want[uncut])
    nseg, ncut, npart = plan.counts()
    empty_first = int((runs[:, 0] == 0).sum()) if nnz else 0       # listed in stage 0 as empty runs (covers_rows)
    assert nseg == int((runs > 0).sum() + (np.ceil(runs / piece)[runs > piece] - 1).sum()) + empty_first
    assert ncut == int((runs > piece).sum()) and npart == int(np.ceil(runs / piece)[runs > piece].sum())
    again = torch.from_numpy(C0).to(dev)
    spmm_cpp.spmm_plan(blk, Bp, again, alpha, beta, spmm_cpp.DevicePlan(blk, nstage, piece, F))
    assert torch.equal(again, Cg)


def test_full_size_reddit_shape_properties():
    """BASELINE config 1 at full size (232,965 nodes, 114.8 M nonzeros) through properties that need no
    CPU oracle: exact linearity under a power-of-two scale, identical columns for identical inputs, row sums
    against an fp64 reduction, planned == plan-less == row kernel on every row no plan cuts, run-to-run bits."""
    import spmm_cpp
    from coo_graph import synthetic, graph_utils
    dev = torch.device('cuda:0')
    n, e, *_ = synthetic.SHAPES['reddit']
    d = synthetic.make_dataset('reddit', seed=0, device=dev, with_features=False)
    adj = graph_utils.sym_normalization(d['edge_index'], n)
    del d
    A = spmm_cpp.CsrBlock.from_coo(adj)
    assert A.n == n and A.nnz == adj._nnz() > 114_000_000
    F = 64
    g = torch.Generator(device=dev).manual_seed(3)
    B = torch.rand(n, F, device=dev, generator=g)
    Z = torch.zeros(n, F, device=dev)
    spmm_cpp.aggregate(A, B, Z)                                   # resident plan (host-built), padded in place
    Z2 = torch.zeros(n, F, device=dev)
    spmm_cpp.aggregate(A, B * 2.0, Z2)
    assert torch.equal(Z2, Z * 2.0)                               # scaling by 2 commutes with every rounding
    Zr = torch.zeros(n, F, device=dev)
    spmm_cpp.aggregate(A, B, Zr)
    assert torch.equal(Zr, Z)                                     # no atomics: same bits every run
    # ones in, row sums out: every column the same bits, and the fp64 row sums within 1e-5
    ones = torch.ones(n, F, device=dev)
    S = torch.zeros(n, F, device=dev)
    spmm_cpp.aggregate(A, ones, S)
    assert torch.equal(S, S[:, :1].expand(-1, F))
    rows = torch.repeat_interleave(torch.arange(n, device=dev), (A.rowptr[1:] - A.rowptr[:-1]).long())
    want = torch.zeros(n, dtype=torch.float64, device=dev).index_add_(0, rows, A.val.double())
    assert ((S[:, 0].double() - want).abs() <= 1e-5 * want.abs()).all()
    del rows, ones, S
    # the three kernels paths agree bit for bit wherever no path cuts the row (runs <= 128 in every plan)
    ci = adj.indices()
    Zc = torch.zeros(n, F, device=dev)
    spmm_cpp.spmm_cusparse(ci[0].int(), ci[1].int(), adj.values(), n, n, B, Zc, 1.0, 1.0)     # plan-less, device-built plan
    Zk = torch.zeros(n, F, device=dev)
    spmm_cpp.spmm_csr(A, B, Zk, 1.0, 1.0)                                                     # row kernel, strict chain
    short = (A.rowptr[1:] - A.rowptr[:-1]) <= 128
    assert int(short.sum()) > 50_000
    assert torch.equal(Z[short], Zk[short]) and torch.equal(Zc[short], Zk[short])
    # cut hub rows (up to 108 K nonzeros): fixed-order piece sums, judged against an fp64 product -- all terms
    # are positive here, so the bound 1e-5 * sum|a*b| is 1e-5 relative, element by element
    exact = torch.sparse.mm(torch.sparse_coo_tensor(ci, adj.values().double(), (n, n)), B.double())
    for got in (Z, Zc):
        assert ((got.double() - exact).abs() <= 1e-5 * exact).all()


_shape_blocks = {}


def _shape_block(shape):
    """The normalised adjacency of a BASELINE config's graph at FULL size as a CsrBlock (+ its COO), built once."""
    import spmm_cpp
    from coo_graph import synthetic, graph_utils
    if shape not in _shape_blocks:
        _shape_blocks.clear()                                     # one big graph resident at a time
        torch.cuda.empty_cache()
        dev = torch.device('cuda:0')
        n = synthetic.SHAPES[shape][0]
        d = synthetic.make_dataset(shape, seed=0, device=dev, with_features=False)
        adj = graph_utils.sym_normalization(d['edge_index'], n)
        del d
        _shape_blocks[shape] = (spmm_cpp.CsrBlock.from_coo(adj), adj)
    return _shape_blocks[shape]


@pytest.mark.parametrize('shape,F', [('reddit', 602), ('reddit', 256), ('reddit', 41),
                                     ('products', 100), ('products', 256), ('products', 47)])
def test_full_size_configs_at_their_own_widths(shape, F):
    """BASELINE configs 1 and 2 at full size AND at the widths their layers aggregate (Reddit 602 / 256 / 41,
    ogbn-products 100 / 256 / 47), through the resident-plan path the models run: bits of the strict row kernel on
    every row the plan does not cut, the fp64 bound (1e-5 relative: all terms are positive) on every element, exact
    linearity under a power-of-two scale, the same bits from run to run, nothing left unwritten by the overwriting
    first stage."""
    import spmm_cpp
    A, adj = _shape_block(shape)
    dev = torch.device('cuda:0')
    n = A.n
    assert A.nnz > (114_000_000 if shape == 'reddit' else 126_000_000)
    g = torch.Generator(device=dev).manual_seed(F)
    B = torch.rand(n, F, device=dev, generator=g)
    Z = torch.full((n, F), float('nan'), device=dev)
    spmm_cpp.aggregate(A, B, Z, overwrite=True)                    # what the first stage of an aggregation runs
    assert torch.isfinite(Z).all()
    plan = spmm_cpp._plan_for(A, F)
    Zk = torch.zeros(n, F, device=dev)
    spmm_cpp.spmm_csr(A, B, Zk, 1.0, 0.0)                          # row kernel: the oracle's chain, element by element
    # rows the plan does not cut into pieces (run plans: every run within the piece length; bundle plans: no bundle the
    # row belongs to is cut): there the result is the strict chain
    uncut = plan.uncut_rows()
    if shape == 'reddit' and F >= 128:
        # row strips take nearly the whole block (tiles of 256 rows share their columns: one B slice per 6-7 nonzeros);
        # every row keeps the strict chain over ascending columns except those of the hub tiles whose union list is cut
        assert isinstance(plan, spmm_cpp.StripHybridPlan) and plan.strip.share > 0.9 and plan.strip.union_ratio < 0.2
        assert int(uncut.sum()) > 0.8 * n
    else:
        assert int(uncut.sum()) > 0.9 * n
    assert torch.equal(Z[uncut], Zk[uncut])
    exact = torch.sparse.mm(torch.sparse_coo_tensor(adj.indices(), adj.values().double(), (n, n)), B.double())
    assert ((Z.double() - exact).abs() <= 1e-5 * exact).all()
    del Zk
    Z2 = torch.zeros(n, F, device=dev)
    spmm_cpp.aggregate(A, B * 2.0, Z2, overwrite=True)
    assert torch.equal(Z2, Z * 2.0)
    spmm_cpp.aggregate(A, B, Z2, overwrite=True)
    assert torch.equal(Z2, Z)
    # accumulate on top (what the later stages of an aggregation do): every chain starts from Z instead of 0. Small
    # terms added one by one to a large running value round more than a sum from 0 does -- in the reference's chain
    # too -- so the yardstick is the strict chain (row kernel, beta = 1): its bits on uncut rows, and on the cut hub
    # rows an error against fp64 no larger than the strict chain's own plus 1e-5.
    spmm_cpp.aggregate(A, B, Z2)
    Zk = Z.clone()
    spmm_cpp.spmm_csr(A, B, Zk, 1.0, 1.0)
    assert torch.equal(Z2[uncut], Zk[uncut])
    assert ((Z2.double() - 2.0 * exact).abs() <= 2e-5 * exact + (Zk.double() - 2.0 * exact).abs()).all()
    del exact, Zk
    if (shape, F) in (('reddit', 41), ('products', 47)):
        _shape_blocks.clear()
        torch.cuda.empty_cache()


@pytest.mark.parametrize('F', [128, 172, 256])
def test_papers_shaped_rank_slab(F):
    """BASELINE config 4's shape through the path an 8-rank job runs: ONE rank's part built on the device without the
    whole graph (synthetic.make_partition, the degree of ogbn-papers100M at 1/28 of its size: 4 M nodes, 58 M entries),
    its row slab multiplied against the gathered features in one SpMM (the all-gather form) and block by block (the
    broadcast form) -- same bits, the row kernel's bits on uncut rows, the fp64 bound everywhere. Widths: input 128,
    hidden 256, 172 classes."""
    import coo_graph
    import spmm_cpp
    from coo_graph import synthetic
    from models import gcn
    dev = torch.device('cuda:0')
    P, rank, n, e = 8, 3, 4_000_000, 29_000_000
    key = ('papers-part', n)
    if key not in _shape_blocks:
        _shape_blocks.clear()
        torch.cuda.empty_cache()
        part = synthetic.make_partition('papers', rank, P, seed=0, device=dev, num_nodes=n, num_edges=e)
        _shape_blocks[key] = coo_graph.Parted_COO_Graph('papers', rank, P, device=dev, attr_dict=part)
    pg = _shape_blocks[key]
    s = pg.local_num_nodes
    assert s == n // P and len(pg.adj_parts) == P and pg.adj_parts.slab.size(1) == s * P
    assert pg.num_classes == 172 and int(pg.local_labels.max()) > 127          # 172 classes do not fit the reference's labels.char()
    g = torch.Generator(device=dev).manual_seed(F)
    H = torch.rand(s * P, F, device=dev, generator=g)
    z_slab = torch.full((s, F), float('nan'), device=dev)
    gcn.spmm(pg.adj_parts.slab, H, z_slab, fresh=True)
    z_blocks = torch.full((s, F), float('nan'), device=dev)
    for src in range(P):
        gcn.spmm(pg.adj_parts[src], H[src * s:(src + 1) * s], z_blocks, fresh=(src == 0))
    slab = gcn._csr_block(pg.adj_parts.slab)
    deg = slab.rowptr[1:] - slab.rowptr[:-1]
    short = deg <= 128
    assert int(short.sum()) > 0.95 * s and int((deg == 1).sum()) > 0          # isolated nodes: the self loop only, nothing in 7 of 8 blocks
    zk = torch.zeros(s, F, device=dev)
    spmm_cpp.spmm_csr(slab, H, zk, 1.0, 0.0)
    assert torch.equal(z_slab[short], zk[short]) and torch.equal(z_blocks[short], zk[short])
    A = pg.adj_parts.slab.coalesce()
    exact = torch.sparse.mm(torch.sparse_coo_tensor(A.indices(), A.values().double(), A.shape), H.double())
    for got in (z_slab, z_blocks):
        assert ((got.double() - exact).abs() <= 1e-5 * exact).all()
    if F == 256:
        _shape_blocks.clear()
        torch.cuda.empty_cache()


def test_papers_rank_slab_at_full_size():
    """BASELINE config 5 at the size it names: rank 3 of 8 of the ogbn-papers100M-shaped R-MAT graph (111 M nodes, 1.6 G
    entries), built on this GPU alone by synthetic.make_partition -- 13.9 M rows, ~200 M nonzeros, gathered operand
    111 M x 128 floats (57 GB) -- through the checks of test_papers_shaped_rank_slab: slab form = block form bit for bit,
    the row kernel's bits on every uncut row, the fp64 bound on a 16-column band at either end (the fp64 operand of all
    128 columns would be 114 GB), run-to-run bits."""
    import coo_graph
    import spmm_cpp
    from coo_graph import synthetic
    from models import gcn
    dev = torch.device('cuda:0')
    _shape_blocks.clear()
    torch.cuda.empty_cache()
    if torch.cuda.mem_get_info()[0] < 200 * 2**30:
        pytest.skip('needs ~150 GB of free HBM')
    P, rank, F = 8, 3, 128
    n = synthetic.SHAPES['papers'][0]
    part = synthetic.make_partition('papers', rank, P, seed=0, device=dev)
    pg = coo_graph.Parted_COO_Graph('papers', rank, P, device=dev, attr_dict=part)
    del part
    s = pg.local_num_nodes
    assert s == (n + P - 1) // P and s > 13_800_000 and pg.adj_parts.slab.size(1) == s * P
    nnz = pg.adj_parts.slab._nnz()
    assert nnz > 150_000_000
    H = torch.rand(s * P, F, device=dev, generator=torch.Generator(device=dev).manual_seed(5))
    z_slab = torch.full((s, F), float('nan'), device=dev)
    gcn.spmm(pg.adj_parts.slab, H, z_slab, fresh=True)
    z_blocks = torch.full((s, F), float('nan'), device=dev)
    for src in range(P):
        gcn.spmm(pg.adj_parts[src], H[src * s:(src + 1) * s], z_blocks, fresh=(src == 0))
    slab = gcn._csr_block(pg.adj_parts.slab)
    deg = slab.rowptr[1:] - slab.rowptr[:-1]
    short = deg <= 128
    assert int(short.sum()) > 0.95 * s and int((deg == 1).sum()) > 0
    zk = torch.zeros(s, F, device=dev)
    spmm_cpp.spmm_csr(slab, H, zk, 1.0, 0.0)
    assert torch.equal(z_slab[short], zk[short]) and torch.equal(z_blocks[short], zk[short])
    del zk
    z2 = torch.full((s, F), float('nan'), device=dev)
    gcn.spmm(pg.adj_parts.slab, H, z2, fresh=True)
    assert torch.equal(z2, z_slab)
    del z2
    A = pg.adj_parts.slab.coalesce()
    A64 = torch.sparse_coo_tensor(A.indices(), A.values().double(), A.shape)
    for c0 in (0, F - 16):
        exact = torch.sparse.mm(A64, H[:, c0:c0 + 16].double())
        for got in (z_slab, z_blocks):
            assert ((got[:, c0:c0 + 16].double() - exact).abs() <= 1e-5 * exact).all()
        del exact
    del A64, A, H, pg, slab
    _shape_blocks.clear()
    torch.cuda.empty_cache()


@pytest.mark.parametrize('n,m,nnz,hubs,F,nstage,piece,hot,alpha,beta', [
    (3000, 5000, 200000, 3, 256, 2, 512, 256, 1.0, 1.0), (2500, 4000, 150000, 2, 100, 3, 256, 512, 0.5, 2.0),
    (1500, 9000, 120000, 2, 602, 4, 1024, 256, 1.0, 0.0), (4000, 1200, 90000, 0, 41, 1, 4096, 512, 1.0, 1.0),
    (700, 700, 30000, 1, 64, 2, 64, 512, -1.0, 1.0)])
def test_hot_tile_plan_matches_the_oracle_on_its_own_entry_order(n, m, nnz, hubs, F, nstage, piece, hot, alpha, beta):
    """Hot-tile plans (north_star's 'feature tiles of H_j staged in LDS'; opt-in, DGNN_HOT_ROWS): per stage the most
    referenced columns' B rows sit in LDS and every run sums its hot entries first. The chain of an output element is
    therefore the plan's own entry order: the CPU oracle run on that order gives the kernel's bits for every row no
    stage cuts; against the true ascending order the result agrees within the fp64 bound like any cut row."""
    import spmm_cpp
    from oracle import ref_cpu
    rng = np.random.default_rng(n + F + hot)
    row, col, val = _rand_csr(rng, n, m, nnz, hub_rows=hubs)
    # a popularity skew, so that the hot set matters: fold half of the columns onto a few
    fold = rng.random(col.size) < 0.5
    col = np.where(fold, col % 97, col).astype(np.int32)
    key = np.unique(row.astype(np.int64) * m + col)                      # re-coalesce
    row, col = (key // m).astype(np.int32), (key % m).astype(np.int32)
    val = rng.random(key.size, dtype=np.float32)
    B = rng.standard_normal((m, F), dtype=np.float32)
    C0 = rng.standard_normal((n, F), dtype=np.float32)
    if beta == 0.0:
        C0[:] = np.nan
    dev = torch.device('cuda:0')
    blk = _gpu_block(row, col, val, n, m)
    plan = spmm_cpp.SegPlan(blk, nstage, piece, hot_rows=hot)
    assert plan.hot_rows == hot and 0.3 < plan.hot_fraction <= 1.0
    Bp = spmm_cpp.pad_rows(torch.from_numpy(B).to(dev))
    Cg = torch.from_numpy(C0).to(dev)
    spmm_cpp.spmm_plan(blk, Bp, Cg, alpha, beta, plan)
    got = Cg.cpu().numpy()
    rowptr = ref_cpu.coo2csr(row, n)
    want_true = ref_cpu.spmm_coo(row, col, val, n, m, B, C0.copy(), alpha, beta)
    _check(got, want_true, rowptr, col, val, B, np.nan_to_num(C0), alpha, beta)      # fp64 bound on every element
    perm = plan.perm.cpu().numpy()
    want_perm = ref_cpu.spmm_csr(rowptr, col[perm], val[perm], B, C0.copy(), alpha, beta)
    cbw = -(-m // nstage)
    runs = np.zeros((n, nstage), dtype=np.int64)
    np.add.at(runs, (row, col // cbw), 1)
    uncut = runs.max(axis=1) <= piece
    assert uncut.sum() > n // 2 and np.array_equal(got[uncut], want_perm[uncut])
    again = torch.from_numpy(C0).to(dev)
    spmm_cpp.spmm_plan(blk, Bp, again, alpha, beta, plan)
    assert torch.equal(again, Cg)


@pytest.mark.parametrize('n,m,nnz,hubs,F,nstage,piece,R,alpha,beta', [
    (3000, 5000, 200000, 3, 256, 2, 512, 4, 1.0, 1.0), (2500, 4000, 150000, 2, 100, 3, 256, 8, 0.5, 2.0),
    (1501, 9000, 120000, 2, 602, 4, 1024, 4, 1.0, 0.0), (4003, 1200, 90000, 0, 41, 1, 4096, 4, 1.0, 1.0),
    (702, 700, 30000, 1, 64, 2, 64, 8, -1.0, 1.0), (9, 40, 60, 0, 128, 1, 16, 4, 1.0, 0.0),
    (2001, 3000, 100000, 2, 256, 3, 512, 6, 1.0, 1.0), (1999, 3000, 80000, 1, 128, 2, 1024, 2, 1.0, 0.0)])
def test_bundle_plan_matches_the_oracle(n, m, nnz, hubs, F, nstage, piece, R, alpha, beta):
    """Row bundles (dgnn_spmm_bundle_f32): R rows share one gather per union column, absent entries weigh 0. Every row
    of an uncut bundle carries the oracle's bits (its own nonzeros in ascending column order); rows of cut bundles meet
    the fp64 bound; beta = 0 overwrites a NaN-filled C completely (row counts that do not divide by R, rows with no
    nonzero in the first column block); same bits from run to run."""
    import spmm_cpp
    from oracle import ref_cpu
    rng = np.random.default_rng(n + F + R)
    row, col, val = _rand_csr(rng, n, m, nnz, hub_rows=hubs)
    fold = rng.random(col.size) < 0.5                                    # popular columns, so that rows share some
    col = np.where(fold, col % 61, col).astype(np.int32)
    key = np.unique(row.astype(np.int64) * m + col)
    row, col = (key // m).astype(np.int32), (key % m).astype(np.int32)
    val = rng.random(key.size, dtype=np.float32) - 0.3
    B = rng.standard_normal((m, F), dtype=np.float32)
    C0 = rng.standard_normal((n, F), dtype=np.float32)
    if beta == 0.0:
        C0[:] = np.nan
    want = ref_cpu.spmm_coo(row, col, val, n, m, B, C0.copy(), alpha, beta)
    dev = torch.device('cuda:0')
    blk = _gpu_block(row, col, val, n, m)
    plan = spmm_cpp.BundlePlan(blk, nstage, R, piece)
    assert plan.union_ratio < 1.0 and plan.R == R
    Bp = spmm_cpp.pad_rows(torch.from_numpy(B).to(dev))
    Cg = torch.from_numpy(C0).to(dev)
    spmm_cpp.spmm_plan(blk, Bp, Cg, alpha, beta, plan)
    got = Cg.cpu().numpy()
    assert np.isfinite(got).all()
    rowptr = ref_cpu.coo2csr(row, n)
    _check(got, want, rowptr, col, val, B, np.nan_to_num(C0), alpha, beta)
    uncut = plan.uncut_rows().cpu().numpy()
    assert (uncut.sum() >= n // 2 or piece < 256) and np.array_equal(got[uncut], want[uncut])     # piece 64: every bundle is cut
    again = torch.from_numpy(C0).to(dev)
    spmm_cpp.spmm_plan(blk, Bp, again, alpha, beta, plan)
    assert torch.equal(again, Cg)
    if beta == 0.0:                                                       # the overwrite entry of the pipeline's first stage
        over = torch.full((n, F), float('nan'), device=dev)
        spmm_cpp.spmm_plan(blk, Bp, over, alpha, 1.0, plan, overwrite=True)
        assert torch.equal(over, Cg)


@pytest.mark.parametrize('n,m,nnz,hubs,F,min_count,min_cols,piece,rest,alpha,beta', [
    (3000, 5000, 250000, 3, 256, 4, 8, 64, 'runs', 1.0, 1.0), (2500, 4000, 200000, 2, 602, 5, 16, 1024, 'bundle', 1.0, 0.0),
    (1501, 9000, 150000, 2, 128, 3, 8, 8, 'runs', 0.5, 2.0), (4003, 1200, 120000, 0, 100, 6, 32, 256, 'bundle', 1.0, 1.0),
    (70, 700, 20000, 1, 130, 3, 8, 1024, 'runs', -1.0, 1.0), (2001, 3000, 160000, 2, 41, 4, 8, 128, 'runs', 1.0, 0.0)])
def test_dense_tiles_plus_remainder_match_the_oracle(n, m, nnz, hubs, F, min_count, min_cols, piece, rest, alpha, beta):
    """Dense tiles (dgnn_spmm_dense_f32, spmm_cpp.DensePlan / HybridPlan): the columns that several of a tile's 32 rows
    share are multiplied as a dense [columns x 32] block (absent entries weigh 0), the other entries go through the
    sparse plan first. Every element meets the fp64 bound; rows in no tile carry the oracle's bits; rows of uncut tiles
    carry the bits of the oracle run on the plan's entry order (remainder ascending, then the listed columns ascending);
    beta = 0 overwrites a NaN-filled C; widths that are no multiple of the 128-float chunk; same bits from run to run."""
    import spmm_cpp
    from oracle import ref_cpu
    rng = np.random.default_rng(n + F + min_count)
    row, col, val = _rand_csr(rng, n, m, nnz, hub_rows=hubs)
    fold = rng.random(col.size) < 0.6                                    # popular columns, so that rows share some
    col = np.where(fold, col % 150, col).astype(np.int32)
    key = np.unique(row.astype(np.int64) * m + col)
    row, col = (key // m).astype(np.int32), (key % m).astype(np.int32)
    val = rng.random(key.size, dtype=np.float32) - 0.3
    B = rng.standard_normal((m, F), dtype=np.float32)
    C0 = rng.standard_normal((n, F), dtype=np.float32)
    if beta == 0.0:
        C0[:] = np.nan
    want = ref_cpu.spmm_coo(row, col, val, n, m, B, C0.copy(), alpha, beta)
    dev = torch.device('cuda:0')
    blk = _gpu_block(row, col, val, n, m)
    D = spmm_cpp.DensePlan(blk, min_count, min_cols, piece, keep_mask=True)
    assert D.nk > 0 and 0.1 < D.share < 1.0 and D.rest.nnz + int(D.in_dense.sum()) == blk.nnz
    rp = spmm_cpp.BundlePlan(D.rest, 2, 4, 256) if rest == 'bundle' else spmm_cpp.SegPlan(D.rest, 2, 256)
    plan = spmm_cpp.HybridPlan(D, rp)
    Bp = spmm_cpp.pad_rows(torch.from_numpy(B).to(dev))
    Cg = torch.from_numpy(C0).to(dev)
    spmm_cpp.spmm_plan(blk, Bp, Cg, alpha, beta, plan)
    got = Cg.cpu().numpy()
    assert np.isfinite(got).all()
    rowptr = ref_cpu.coo2csr(row, n)
    _check(got, want, rowptr, col, val, B, np.nan_to_num(C0), alpha, beta)               # fp64 bound on every element
    strict = plan.strict_rows().cpu().numpy()
    assert np.array_equal(got[strict], want[strict])
    in_dense = D.in_dense.cpu().numpy()
    perm = np.argsort(row.astype(np.int64) * 2 + in_dense, kind='stable')                # per row: remainder first, then the tiles' entries
    want_perm = ref_cpu.spmm_csr(rowptr, col[perm], val[perm], B, C0.copy(), alpha, beta)
    uncut = plan.uncut_rows().cpu().numpy()
    touched = D.touched_rows().cpu().numpy()
    assert (uncut & touched).sum() > 0 or piece <= 8
    assert np.array_equal(got[uncut], want_perm[uncut])
    again = torch.from_numpy(C0).to(dev)
    spmm_cpp.spmm_plan(blk, Bp, again, alpha, beta, plan)
    assert torch.equal(again, Cg)
    if beta == 0.0:
        over = torch.full((n, F), float('nan'), device=dev)
        spmm_cpp.spmm_plan(blk, Bp, over, alpha, 1.0, plan, overwrite=True)
        assert torch.equal(over, Cg)
    # a C whose leading dimension is wider than F and not a multiple of 4 (scalar / float2 stores at the edge)
    wide = torch.full((n, F + 3), 7.0, device=dev)
    view = wide[:, :F]
    view.copy_(torch.from_numpy(np.nan_to_num(C0)).to(dev))
    spmm_cpp.spmm_plan(blk, Bp, view, alpha, 0.0 if beta == 0.0 else beta, plan)
    assert torch.equal(view, Cg) and (wide[:, F:] == 7.0).all()


@pytest.mark.parametrize('n,m,nnz,hubs,F,max_ratio,piece,block,beta', [
    (3000, 5000, 250000, 3, 256, 0.9, 256, 26, 1.0), (2500, 4000, 200000, 2, 602, 0.9, 8, 26, 0.0),
    (1501, 9000, 150000, 2, 128, 0.95, 3, 10, 2.0), (4003, 1200, 120000, 0, 100, 0.9, 256, 60, 1.0),
    (700, 700, 60000, 4, 130, 1.0, 256, 26, 1.0), (2001, 3000, 160000, 2, 41, 0.9, 5, 26, 0.0),
    (5000, 6000, 400000, 6, 300, 0.6, 16, 200, 1.0)])
def test_row_strips_plus_rest_match_the_oracle(n, m, nnz, hubs, F, max_ratio, piece, block, beta):
    """Row strips (dgnn_spmm_strip_f32, spmm_cpp.StripPlan / StripHybridPlan): tiles of 256 rows whose union column list
    goes through LDS once, a flat entry list per wave multiplied in VGPR index mode; tiles with poor reuse stay with the
    sparse plans. A row is whole in one of the two parts and sums its nonzeros in ascending column order, so EVERY row of
    an uncut tile carries the oracle's bits -- hub rows whose blocks span several entry lines, strips of 8 to 64 columns,
    widths that are no multiple of the 256-float chunk, rows with nothing; rows of cut tiles meet the fp64 bound; beta = 0
    overwrites a NaN-filled C; pad entries change nothing; same bits from run to run."""
    import spmm_cpp
    from oracle import ref_cpu
    rng = np.random.default_rng(n + F + piece)
    row, col, val = _rand_csr(rng, n, m, nnz, hub_rows=hubs)
    fold = rng.random(col.size) < 0.6                                    # popular columns, so that rows share some
    col = np.where(fold, col % 150, col).astype(np.int32)
    key = np.unique(row.astype(np.int64) * m + col)
    row, col = (key // m).astype(np.int32), (key % m).astype(np.int32)
    val = rng.random(key.size, dtype=np.float32) - 0.3
    B = rng.standard_normal((m, F), dtype=np.float32)
    C0 = rng.standard_normal((n, F), dtype=np.float32)
    if beta == 0.0:
        C0[:] = np.nan
    want = ref_cpu.spmm_coo(row, col, val, n, m, B, C0.copy(), 1.0, beta)
    dev = torch.device('cuda:0')
    blk = _gpu_block(row, col, val, n, m)
    S = spmm_cpp.StripPlan(blk, max_ratio, piece, min_tile_nnz=64, block_target=block)
    assert S.npiece > 0 and 0 < S.share <= 1.0 and S.union_ratio <= max_ratio and S.rest.nnz == key.size - round(S.share * key.size)
    widths = set(S.desc[:, 5].cpu().tolist())
    assert widths <= {8, 16, 32, 64}
    plan = spmm_cpp.StripHybridPlan(S, spmm_cpp.SegPlan(S.rest, 2))
    Bp = spmm_cpp.pad_rows(torch.from_numpy(B).to(dev))
    Cg = torch.from_numpy(C0).to(dev)
    spmm_cpp.spmm_plan(blk, Bp, Cg, 1.0, beta, plan)
    got = Cg.cpu().numpy()
    assert np.isfinite(got).all()
    rowptr = ref_cpu.coo2csr(row, n)
    _check(got, want, rowptr, col, val, B, np.nan_to_num(C0), 1.0, beta)
    uncut = plan.uncut_rows().cpu().numpy()
    in_strips = S.strip_rows().cpu().numpy()
    assert in_strips.sum() >= 256 and (piece < 64 or (uncut & in_strips).sum() >= 256)
    assert np.array_equal(got[uncut], want[uncut])                        # the strict chain's bits, strip rows included
    if piece < 64:
        assert S.ncut > 0 and S.npartial > 0                              # (these cases cut tiles into pieces)
    again = torch.from_numpy(C0).to(dev)
    spmm_cpp.spmm_plan(blk, Bp, again, 1.0, beta, plan)
    assert torch.equal(again, Cg)
    if beta == 0.0:
        over = torch.full((n, F), float('nan'), device=dev)
        spmm_cpp.spmm_plan(blk, Bp, over, 1.0, 1.0, plan, overwrite=True)
        assert torch.equal(over, Cg)
    # Every row written once (dgnn_spmm_strip_overwrite_f32 + a plan of the rest that skips the tiles' rows): a NaN-filled C,
    # neither read nor zero-filled, ends up with the bits of "zero-fill, rest, then add the strips" -- signs of zeros included
    base = torch.zeros(n, F, device=dev)
    spmm_cpp.spmm_plan(blk, Bp, base, 1.0, 0.0, plan)
    fresh = spmm_cpp.SegPlan(S.rest, 2, skip_rows=S.strip_rows())
    assert fresh.nseg == (plan.rest_plan.nseg if S.rest.nnz else n) - int(in_strips.sum()) and fresh.ncut == plan.rest_plan.ncut
    once = spmm_cpp.StripHybridPlan(S, plan.rest_plan, fresh)
    over = torch.full((n, F), float('nan'), device=dev)
    spmm_cpp.spmm_plan(blk, Bp, over, 1.0, 1.0, once, overwrite=True)
    assert torch.equal(over.view(torch.int32), base.view(torch.int32))
    if S.rest.nnz:
        with pytest.raises(ValueError, match='skip_rows'):
            spmm_cpp.SegPlan(S.rest, 2, skip_rows=~S.strip_rows())      # rows that hold entries cannot be skipped
    with pytest.raises(RuntimeError, match='alpha'):
        spmm_cpp.spmm_plan(blk, Bp, Cg, 0.5, 1.0, plan)


def test_row_strips_alone_write_a_block_whose_tiles_are_all_selected():
    """Every tile selected: nothing is left for the plan of the rest but the zeros of rows outside the tiles -- there are
    none here -- and the overwriting product is the strip call alone plus an empty plan."""
    import spmm_cpp
    from oracle import ref_cpu
    rng = np.random.default_rng(5)
    n, m, F = 512, 300, 200
    row, col, val = _rand_csr(rng, n, m, 40000, hub_rows=0)
    B = rng.standard_normal((m, F), dtype=np.float32)
    dev = torch.device('cuda:0')
    blk = _gpu_block(row, col, val, n, m)
    S = spmm_cpp.StripPlan(blk, 1.0, 100000, min_tile_nnz=1)             # (no tile cut: every row one chain)
    assert S.rest.nnz == 0 and S.share == 1.0 and bool(S.strip_rows().all()) and S.ncut == 0
    rest_plan = spmm_cpp.SegPlan(S.rest, 1)
    fresh = spmm_cpp.SegPlan(S.rest, 1, skip_rows=S.strip_rows())
    assert fresh.nseg == 0
    Bp = spmm_cpp.pad_rows(torch.from_numpy(B).to(dev))
    want = ref_cpu.spmm_coo(row, col, val, n, m, B, np.zeros((n, F), dtype=np.float32), 1.0, 0.0)
    for hp in (spmm_cpp.StripHybridPlan(S, rest_plan), spmm_cpp.StripHybridPlan(S, rest_plan, fresh)):
        C = torch.full((n, F), float('nan'), device=dev)
        spmm_cpp.spmm_plan(blk, Bp, C, 1.0, 1.0, hp, overwrite=True)
        assert np.array_equal(C.cpu().numpy(), want)


def test_row_strips_keep_non_finite_values_with_their_neighbours():
    """A non-finite feature row reaches exactly the rows that reference it (the reference's SpMM, spmm.cpp:26-28, has no
    other behaviour): the strip kernel's pad entries multiply an all-zero slice by -0.0 and add nothing, and no row is
    given a zero weight for a column it does not own. Signed zeros survive too: a row of negative zeros stays negative."""
    import spmm_cpp
    rng = np.random.default_rng(77)
    n = m = 1500
    row, col, val = _rand_csr(rng, n, m, 120000, hub_rows=2)
    col = np.where(rng.random(col.size) < 0.6, col % 100, col).astype(np.int32)
    key = np.unique(row.astype(np.int64) * m + col)
    row, col = (key // m).astype(np.int32), (key % m).astype(np.int32)
    val = rng.random(key.size, dtype=np.float32) + 0.1
    F = 256
    B = rng.standard_normal((m, F), dtype=np.float32)
    bad_inf, bad_nan = 7, 1203
    B[bad_inf] = np.inf
    B[bad_nan] = np.nan
    dev = torch.device('cuda:0')
    blk = _gpu_block(row, col, val, n, m)
    S = spmm_cpp.StripPlan(blk, 1.0, 256, min_tile_nnz=64)
    assert S.share > 0.5
    plan = spmm_cpp.StripHybridPlan(S, spmm_cpp.SegPlan(S.rest, 1))
    Cg = torch.full((n, F), float('nan'), device=dev)
    spmm_cpp.spmm_plan(blk, spmm_cpp.pad_rows(torch.from_numpy(B).to(dev)), Cg, 1.0, 0.0, plan)
    got = Cg.cpu().numpy()
    touched = np.zeros(n, dtype=bool)
    touched[row[(col == bad_inf) | (col == bad_nan)]] = True
    assert touched.any() and not touched.all()
    assert np.isfinite(got[~touched]).all() and not np.isfinite(got[touched]).all(axis=1).any()
    # signed zeros: C = -0.0 and a matrix whose products are all -0.0 or absent
    Bz = np.zeros((m, F), dtype=np.float32)
    Cz = torch.full((n, F), -0.0, device=dev)
    neg = _gpu_block(row, col, -np.abs(val), n, m)
    Sz = spmm_cpp.StripPlan(neg, 1.0, 256, min_tile_nnz=64)
    spmm_cpp.spmm_plan(neg, spmm_cpp.pad_rows(torch.from_numpy(Bz).to(dev)), Cz, 1.0, 1.0, spmm_cpp.StripHybridPlan(Sz, spmm_cpp.SegPlan(Sz.rest, 1)))
    assert bool(torch.signbit(Cz).all())                                   # -0.0 + (-a) * (+0.0) = -0.0, pads included


def test_strict_switch_gives_every_row_the_oracle_bits(monkeypatch):
    """DGNN_STRICT=1 (parity / debugging runs): no strips, tiles or bundles and no cut runs -- every row, hub rows included,
    carries the CPU oracle's bits, and a non-finite feature stays with its neighbours."""
    import spmm_cpp
    from oracle import ref_cpu
    rng = np.random.default_rng(5)
    n, m, F = 3000, 3000, 256
    row, col, val = _rand_csr(rng, n, m, 300000, hub_rows=4)
    col = np.where(rng.random(col.size) < 0.6, col % 120, col).astype(np.int32)
    key = np.unique(row.astype(np.int64) * m + col)
    row, col = (key // m).astype(np.int32), (key % m).astype(np.int32)
    val = rng.random(key.size, dtype=np.float32) - 0.3
    B = rng.standard_normal((m, F), dtype=np.float32)
    B[17] = np.inf
    want = ref_cpu.spmm_coo(row, col, val, n, m, B, np.zeros((n, F), dtype=np.float32), 1.0, 0.0)
    blk = _gpu_block(row, col, val, n, m)
    for k in ('DGNN_DENSE_MIN_NNZ', 'DGNN_STRIP_MIN_NNZ'):
        monkeypatch.setenv(k, '1')
    monkeypatch.setenv('DGNN_DENSE_MIN_DEG', '1')
    monkeypatch.setenv('DGNN_STRIP_MIN_DEG', '1')
    fast = spmm_cpp._plan_for(blk, F)
    assert isinstance(fast, (spmm_cpp.StripHybridPlan, spmm_cpp.HybridPlan, spmm_cpp.BundlePlan))
    monkeypatch.setenv('DGNN_STRICT', '1')
    plan = spmm_cpp._plan_for(blk, F)
    assert isinstance(plan, spmm_cpp.SegPlan) and plan.ncut == 0
    dev = torch.device('cuda:0')
    Cg = torch.full((n, F), float('nan'), device=dev)
    spmm_cpp.spmm_plan(blk, spmm_cpp.pad_rows(torch.from_numpy(B).to(dev)), Cg, 1.0, 0.0, plan)
    got = Cg.cpu().numpy()
    assert np.array_equal(got, want, equal_nan=True)
    touched = np.zeros(n, dtype=bool)
    touched[row[col == 17]] = True
    assert np.isfinite(got[~touched]).all() and touched.any()


def test_dense_entry_point_rejects_bad_arguments():
    import spmm_cpp
    lib = spmm_cpp.lib
    lib.dgnn_last_error.restype = ctypes.c_char_p
    rc = lib.dgnn_spmm_dense_f32(None, 1, 1, None, ctypes.c_int64(4), None, ctypes.c_int64(4), None, ctypes.c_int64(4), 4,
                                 ctypes.c_float(1), None)
    assert rc == 1 and b'null plan' in lib.dgnn_last_error()
    dev = torch.device('cuda:0')
    rng = np.random.default_rng(5)
    row, col, val = _rand_csr(rng, 200, 300, 20000)
    blk = _gpu_block(row, col, val, 200, 300)
    D = spmm_cpp.DensePlan(blk, 3, 8, 64)
    assert D.nk > 0
    B = torch.rand(300, 130, device=dev)                                     # rows not 16-byte aligned: refused
    C = torch.zeros(200, 130, device=dev)
    with pytest.raises(RuntimeError, match='ldb % 4'):
        spmm_cpp.spmm_dense(D, B, C)
    with pytest.raises(ValueError):
        spmm_cpp.DensePlan(blk, 3, 8, 100)                                   # piece length must be a multiple of 8


def test_row_bucketed_order_is_chosen_for_blocks_with_locality_and_changes_no_bit():
    """A graph with communities, relabelled by the cluster order (prepare_data.py --reorder cluster): the plan rule
    sees the locality (spmm_cpp.locality_share) and lists the runs longest-first inside buckets of 4096 consecutive
    rows, in one stage. Only the ORDER of the work changes: the result keeps the strict chain's bits on every uncut
    row, as with the global order; a scrambled graph keeps the global order."""
    import spmm_cpp
    from coo_graph import graph_utils, reorder, synthetic
    dev = torch.device('cuda:0')
    n = 70000
    keys = synthetic.planted_undirected(n, 700000, community=2048, seed=4, device=dev)
    lo, hi = keys // n, keys % n
    ei = torch.stack([torch.cat([lo, hi]), torch.cat([hi, lo])])
    order = reorder.cluster_order(ei, n)
    new = torch.empty(n, dtype=torch.long, device=dev)
    new[order] = torch.arange(n, device=dev)
    B = torch.rand(n, 200, device=dev)
    for edges, local in ((new[ei], True), (ei, False)):
        A = spmm_cpp.CsrBlock.from_coo(graph_utils.sym_normalization(edges, n))
        assert A.nnz >= 1 << 20
        plan = spmm_cpp._plan_for(A, 200)
        assert isinstance(plan, spmm_cpp.SegPlan)
        assert (plan.row_bucket == spmm_cpp.ROW_BUCKET and plan.nstage == 1) if local else plan.row_bucket == 0
        if local:                                              # inside a bucket: longest first; buckets in row order
            seg = plan.seg.long()
            bk = seg[:, 0] // spmm_cpp.ROW_BUCKET
            assert bool((bk[1:] >= bk[:-1]).all())
            ln = seg[:, 2] - seg[:, 1]
            same = bk[1:] == bk[:-1]
            assert bool((ln[1:][same] <= ln[:-1][same]).all())
        Z = torch.full((n, 200), float('nan'), device=dev)
        spmm_cpp.aggregate(A, B, Z, overwrite=True)
        Zk = torch.zeros(n, 200, device=dev)
        spmm_cpp.spmm_csr(A, B, Zk, 1.0, 0.0)
        uncut = plan.uncut_rows()
        assert int(uncut.sum()) > 0.99 * n and torch.equal(Z[uncut], Zk[uncut])
        assert ((Z - Zk).abs() <= 1e-5 * Zk.abs()).all()


def test_plan_builders_refuse_a_block_with_repeated_entries():
    """Tiles, bundles and strips store ONE value per (row, column): a hand-built block that lists an entry twice (never a
    coalesced one) would silently lose a value -- the builders refuse it; the run plans and the row kernel add both."""
    import spmm_cpp
    dev = torch.device('cuda:0')
    rowptr = torch.tensor([0, 3, 4], dtype=torch.int32, device=dev)
    col = torch.tensor([1, 1, 2, 0], dtype=torch.int32, device=dev)
    val = torch.tensor([1.0, 2.0, 3.0, 4.0], device=dev)
    A = spmm_cpp.CsrBlock(2, 3, rowptr, col, val)
    for build in (lambda: spmm_cpp.BundlePlan(A), lambda: spmm_cpp.DensePlan(A), lambda: spmm_cpp.StripPlan(A)):
        with pytest.raises(RuntimeError, match='more than once'):
            build()
    B = spmm_cpp.pad_rows(torch.eye(3, device=dev))
    C = torch.zeros(2, 3, device=dev)
    spmm_cpp.spmm_csr(A, B[:, :3], C, 1.0, 0.0)
    assert C.tolist() == [[0.0, 3.0, 3.0], [4.0, 0.0, 0.0]]
