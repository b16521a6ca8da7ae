"""Randomised parity of every SpMM entry form against the CPU oracle (hypothesis, derandomised so that a run is
repeatable): shapes down to one row / one column / one float, empty rows, hub rows, any stage count and piece
length, strided B and C, any alpha and beta. Uncut rows must carry the oracle's bits; cut rows the fp64 bound."""
import numpy as np
import pytest
import torch
from hypothesis import HealthCheck, given, settings, strategies as st

from test_spmm_gpu import _check, _gpu_block, _rand_csr

pytestmark = pytest.mark.gpu


@st.composite
def _cases(draw):
    n = draw(st.integers(1, 700))
    m = draw(st.integers(1, 900))
    density = draw(st.sampled_from([0.0, 0.002, 0.02, 0.2]))
    F = draw(st.one_of(st.integers(1, 70), st.sampled_from([100, 128, 172, 256, 300, 602])))
    return dict(n=n, m=m, nnz=int(n * m * density), F=F,
                hubs=draw(st.integers(0, 2)), nstage=draw(st.integers(1, 7)),
                piece=draw(st.sampled_from([1, 5, 16, 33, 256, 1000, 100000])),
                alpha=draw(st.sampled_from([1.0, 1.0, 0.5, -2.0])), beta=draw(st.sampled_from([1.0, 1.0, 0.0, 0.25])),
                pad_b=draw(st.sampled_from([0, 4, 32])), pad_c=draw(st.sampled_from([0, 1, 2, 4])),
                builder=draw(st.sampled_from(['torch', 'device'])), seed=draw(st.integers(0, 2 ** 20)))


@settings(max_examples=400, deadline=None, derandomize=True, suppress_health_check=list(HealthCheck))
@given(_cases())
def test_every_entry_form_matches_the_oracle(c):
    import spmm_cpp
    from oracle import ref_cpu
    rng = np.random.default_rng(c['seed'])
    n, m, F = c['n'], c['m'], c['F']
    row, col, val = _rand_csr(rng, n, m, c['nnz'], hub_rows=c['hubs'] if m >= 8 else 0)
    B = rng.standard_normal((m, F), dtype=np.float32)
    C0 = rng.standard_normal((n, F), dtype=np.float32)
    if c['beta'] == 0.0:
        C0[:] = np.nan
    want = ref_cpu.spmm_coo(row, col, val, n, m, B, C0.copy(), c['alpha'], c['beta'])
    rowptr = ref_cpu.coo2csr(row, n)
    dev = torch.device('cuda:0')
    blk = _gpu_block(row, col, val, n, m)
    Bd = torch.from_numpy(B).to(dev)

    def fresh_c():                                     # C as a view of a wider buffer: any leading dimension
        buf = torch.full((n, F + c['pad_c']), 7.0, device=dev)
        buf[:, :F] = torch.from_numpy(C0).to(dev)
        return buf, buf[:, :F]

    # 1. general kernel, CSR operands, B at any pitch
    Bw = torch.zeros(m, F + c['pad_b'], device=dev)
    Bw[:, :F] = Bd
    buf, C = fresh_c()
    spmm_cpp.spmm_csr(blk, Bw[:, :F], C, c['alpha'], c['beta'])
    assert np.array_equal(C.cpu().numpy(), want), 'row kernel must carry the oracle chain'
    assert (buf[:, F:] == 7.0).all()

    # 2. the reference's call (COO, no plan)
    buf, C = fresh_c()
    spmm_cpp.spmm_cusparse(torch.from_numpy(row).to(dev), torch.from_numpy(col).to(dev), torch.from_numpy(val).to(dev),
                           n, m, Bd, C, c['alpha'], c['beta'])
    _check(C.cpu().numpy(), want, rowptr, col, val, B, np.nan_to_num(C0), c['alpha'], c['beta'])
    assert (buf[:, F:] == 7.0).all()

    # 3. fast path with a plan (host- or device-built), padded B
    plan = (spmm_cpp.SegPlan(blk, c['nstage'], c['piece']) if c['builder'] == 'torch'
            else spmm_cpp.DevicePlan(blk, min(c['nstage'], 16), c['piece'], F))
    Bp = spmm_cpp.pad_rows(Bd)
    buf, C = fresh_c()
    spmm_cpp.spmm_plan(blk, Bp, C, c['alpha'], c['beta'], plan)
    got = C.cpu().numpy()
    _check(got, want, rowptr, col, val, B, np.nan_to_num(C0), c['alpha'], c['beta'])
    assert (buf[:, F:] == 7.0).all(), 'nothing may be written past column F'
    lens = np.diff(rowptr)
    whole = lens <= c['piece']                         # rows no plan can have cut
    assert np.array_equal(got[whole], want[whole])

    # 4. the overwriting product on an uninitialised (NaN) C equals beta = 0
    if c['beta'] == 0.0:
        buf = torch.full((n, F + c['pad_c']), float('nan'), device=dev)
        spmm_cpp.spmm_plan(blk, Bp, buf[:, :F], c['alpha'], 1.0, plan, overwrite=True)
        assert np.array_equal(buf[:, :F].cpu().numpy(), got) and torch.isnan(buf[:, F:]).all()


@settings(max_examples=150, deadline=None, derandomize=True, suppress_health_check=list(HealthCheck))
@given(n=st.integers(1, 400), F=st.integers(1, 700), src_pad=st.integers(0, 5), extra=st.sampled_from([0, 4, 32]),
       k=st.integers(0, 500), off=st.integers(0, 3), seed=st.integers(0, 2 ** 20))
def test_pad_and_gather_rows_are_exact_copies(n, F, src_pad, extra, k, off, seed):
    """dgnn_pad_rows_f32 / dgnn_gather_rows_f32 for any source pitch and alignment: the first F columns are the
    source's bits, columns F..ldd-1 are zero, rows beyond k and anything outside the destination are untouched."""
    import spmm_cpp
    dev = torch.device('cuda:0')
    g = torch.Generator(device=dev).manual_seed(seed)
    flat = torch.rand(n * (F + src_pad) + off, device=dev, generator=g)
    src = flat[off:].view(n, F + src_pad)[:, :F]            # any pitch, base possibly only 4-byte aligned
    ldd = spmm_cpp.padded_width(F) + extra
    dst = torch.full((n + 1, ldd), 9.0, device=dev)
    out = spmm_cpp.pad_rows(src, dst[:n])
    assert out.data_ptr() == dst.data_ptr()
    assert torch.equal(dst[:n, :F], src) and (dst[:n, F:] == 0).all() and (dst[n] == 9.0).all()
    idx = torch.randint(0, n, (k,), device=dev, generator=g).int()
    dst2 = torch.full((k + 1, ldd), 9.0, device=dev)
    spmm_cpp.gather_rows(src, idx, dst2[:k] if k else dst2[:0])
    assert torch.equal(dst2[:k, :F], src[idx.long()]) and (dst2[:k, F:] == 0).all() and (dst2[k] == 9.0).all()


@settings(max_examples=120, deadline=None, derandomize=True, suppress_health_check=list(HealthCheck))
@given(n=st.integers(1, 3000), m=st.integers(1, 3000), density=st.sampled_from([0.0, 0.001, 0.01, 0.1]),
       hubs=st.integers(0, 3), nstage=st.integers(1, 16), piece=st.sampled_from([1, 7, 64, 500, 4096]),
       seed=st.integers(0, 2 ** 20))
def test_device_built_plan_covers_every_nonzero_once(n, m, density, hubs, nstage, piece, seed):
    """dgnn_plan_build: the live segments tile the nonzeros exactly once, stay inside one column block and one
    row, respect the piece length; cut lists and partial-row numbering are consistent; counters match."""
    import spmm_cpp
    rng = np.random.default_rng(seed)
    row, col, val = _rand_csr(rng, n, m, int(n * m * density), hub_rows=hubs if m >= 8 else 0)
    blk = _gpu_block(row, col, val, n, m)
    plan = spmm_cpp.DevicePlan(blk, nstage, piece, 64)
    nseg, ncut, npart = plan.counts()
    if blk.nnz == 0:
        assert (nseg, ncut, npart) == (0, 0, 0)
        return
    lay = plan.layout
    ws = plan.ws.cpu().numpy()
    seg = ws[lay.hdr_ints:lay.hdr_ints + 4 * lay.nstage * lay.nseg_max].reshape(lay.nstage, lay.nseg_max, 4)
    cut = ws[lay.hdr_ints + 4 * lay.nstage * lay.nseg_max:][:4 * lay.nstage * lay.ncut_max].reshape(lay.nstage, lay.ncut_max, 4)
    cbw = max(1, -(-m // lay.nstage))
    covered = np.zeros(blk.nnz, dtype=np.int32)
    live_total, pieces_seen = 0, {}
    for st_ in range(lay.nstage):
        live = seg[st_][seg[st_][:, 0] >= 0]
        live_total += len(live)
        for r, k0, k1, out in live:
            assert 0 <= k1 - k0 <= lay.piece and (row[k0:k1] == r).all()
            assert k1 > k0 or (st_ == 0 and out < 0)          # empty runs: only in stage 0 (covers_rows)
            assert (np.minimum(col[k0:k1] // cbw, lay.nstage - 1) == st_).all()
            covered[k0:k1] += 1
            if out >= 0:
                pieces_seen[out] = pieces_seen.get(out, 0) + 1
        for r, pbase, cnt, _ in cut[st_][cut[st_][:, 2] > 0]:
            assert all(pieces_seen.get(p, 0) == 1 for p in range(pbase, pbase + cnt))
    assert (covered == 1).all() and live_total == nseg
    assert plan.c.covers_rows == 1
    cut_rows0 = {int(r_) for r_, pb, cnt, _ in cut[0][cut[0][:, 2] > 0]}
    seen0 = np.zeros(n, dtype=np.int32)
    for r, k0, k1, out in seg[0][seg[0][:, 0] >= 0]:
        if out < 0:
            seen0[r] += 1
    for r in cut_rows0:
        seen0[r] += 1
    assert (seen0 == 1).all(), 'the first stage must store every row exactly once'
    assert sum(pieces_seen.values()) == npart == len(pieces_seen) and npart <= lay.npartial_max
    assert ncut == int((cut[:, :, 2] > 0).sum())


@st.composite
def _shared_column_cases(draw):
    n = draw(st.integers(1, 600))
    m = draw(st.integers(8, 900))
    return dict(n=n, m=m, nnz=int(n * m * draw(st.sampled_from([0.01, 0.05, 0.2]))),
                F=draw(st.one_of(st.integers(1, 70), st.sampled_from([100, 128, 130, 172, 256, 300, 602]))),
                fold=draw(st.sampled_from([0.0, 0.5, 0.9])), popular=draw(st.integers(1, 200)),
                hubs=draw(st.integers(0, 2)), nstage=draw(st.integers(1, 5)),
                piece=draw(st.sampled_from([8, 16, 64, 256, 4096])), R=draw(st.sampled_from([2, 4, 6, 8])),
                min_count=draw(st.integers(1, 8)), min_cols=draw(st.sampled_from([1, 8, 40])),
                rest=draw(st.sampled_from(['runs', 'bundle'])),
                alpha=draw(st.sampled_from([1.0, 1.0, 0.5, -2.0])), beta=draw(st.sampled_from([1.0, 1.0, 0.0, 0.25])),
                pad_c=draw(st.sampled_from([0, 1, 2, 4])), seed=draw(st.integers(0, 2 ** 20)))


@settings(max_examples=200, deadline=None, derandomize=True, suppress_health_check=list(HealthCheck))
@given(_shared_column_cases())
def test_bundle_and_dense_tile_forms_match_the_oracle(c):
    """Row bundles and dense tiles + remainder on random blocks whose rows share columns to a random degree: the fp64
    bound on every element, the oracle's bits on every row the plan leaves whole (bundles: rows of uncut bundles;
    tiles: rows in no tile), nothing written past column F, beta = 0 never reads C."""
    import spmm_cpp
    from oracle import ref_cpu
    rng = np.random.default_rng(c['seed'])
    n, m, F = c['n'], c['m'], c['F']
    row, col, val = _rand_csr(rng, n, m, c['nnz'], hub_rows=c['hubs'])
    if c['fold'] and row.size:
        fold = rng.random(col.size) < c['fold']
        col = np.where(fold, col % min(c['popular'], m), col).astype(np.int32)
        key = np.unique(row.astype(np.int64) * m + col)
        row, col = (key // m).astype(np.int32), (key % m).astype(np.int32)
        val = rng.random(key.size, dtype=np.float32) - 0.3
    B = rng.standard_normal((m, F), dtype=np.float32)
    C0 = rng.standard_normal((n, F), dtype=np.float32)
    if c['beta'] == 0.0:
        C0[:] = np.nan
    want = ref_cpu.spmm_coo(row, col, val, n, m, B, C0.copy(), c['alpha'], c['beta'])
    rowptr = ref_cpu.coo2csr(row, n)
    dev = torch.device('cuda:0')
    blk = _gpu_block(row, col, val, n, m)
    Bp = spmm_cpp.pad_rows(torch.from_numpy(B).to(dev))

    def run(plan):
        buf = torch.full((n, F + c['pad_c']), 7.0, device=dev)
        buf[:, :F] = torch.from_numpy(C0).to(dev)
        spmm_cpp.spmm_plan(blk, Bp, buf[:, :F], c['alpha'], c['beta'], plan)
        assert (buf[:, F:] == 7.0).all(), 'nothing may be written past column F'
        got = buf[:, :F].cpu().numpy()
        _check(got, want, rowptr, col, val, B, np.nan_to_num(C0), c['alpha'], c['beta'])
        return got

    bundle = spmm_cpp.BundlePlan(blk, c['nstage'], c['R'], c['piece'])
    got = run(bundle)
    whole = bundle.uncut_rows().cpu().numpy()
    assert np.array_equal(got[whole], want[whole])

    D = spmm_cpp.DensePlan(blk, c['min_count'], c['min_cols'], c['piece'])
    rest = (spmm_cpp.BundlePlan(D.rest, c['nstage'], 4, max(c['piece'], 16)) if c['rest'] == 'bundle'
            else spmm_cpp.SegPlan(D.rest, c['nstage'], c['piece']))
    hybrid = spmm_cpp.HybridPlan(D, rest)
    got = run(hybrid)
    strict = hybrid.strict_rows().cpu().numpy()
    assert np.array_equal(got[strict], want[strict])
    assert D.rest.nnz + round(D.share * blk.nnz) == blk.nnz
