"""spmm_cpp.StripPlan, the host-side builder of struct dgnn_strip_plan (include/dgnn_spmm.h), on the CPU: the plan is decoded
the way spmm_strip_kernel walks it -- pieces, waves, strips, lines, groups of 4 -- and must give back every nonzero of the
selected tiles exactly once, in (row, column) order inside a block, with pad entries that change nothing, the column ids of
strip s + 2 in the first line of strip s, and the unselected rows whole in `rest`. Both ways of finding a tile's union
(bitmap; sort, for blocks with too many (tile, column) cells) give the same plan."""
import numpy as np
import pytest
import torch


def _block(n, m, nnz, hubs, seed):
    import spmm_cpp
    rng = np.random.default_rng(seed)
    row = rng.integers(0, n, nnz)
    col = np.where(rng.random(nnz) < 0.6, rng.integers(0, max(m // 20, 1), nnz), rng.integers(0, m, nnz))
    row[:hubs * 40] = rng.integers(0, max(hubs, 1), hubs * 40)             # a few long rows
    key = np.unique(row.astype(np.int64) * m + col)
    row, col = (key // m).astype(np.int32), (key % m).astype(np.int32)
    val = rng.standard_normal(key.size).astype(np.float32)
    rp = np.zeros(n + 1, dtype=np.int32)
    np.cumsum(np.bincount(row, minlength=n), out=rp[1:])
    return spmm_cpp.CsrBlock(n, m, torch.from_numpy(rp), torch.from_numpy(col), torch.from_numpy(val)), (row, col, val)


def _decode(S):
    """Every (row, column, value bits) the kernel would multiply, walking the plan as spmm_strip_kernel does."""
    desc, rows, ucol, ebase = S.desc.numpy(), S.rows.numpy(), S.ucol.numpy(), S.ebase.numpy()
    ent = S.ent.numpy().reshape(-1, 2, 32)                                  # line, {meta, value}, entry
    out = []
    for p in range(S.npiece):
        tile, k0, k1, _, nsp, Wt = (int(x) for x in desc[p][:6])
        assert Wt in (8, 16, 32, 64) and 0 < nsp and k0 < k1
        for w in range(16):
            line = int(ebase[p, w])
            for sp in range(nsp):
                first, prev = True, (-1, -1)
                while True:
                    meta, vals = ent[line, 0].view(np.uint32), ent[line, 1]
                    ng, last = (int(meta[0]) >> 20) & 15, (int(meta[0]) >> 24) & 1
                    assert 1 <= ng <= (7 if first else 8)
                    if first:                                               # column ids of the wave's slots of strip sp + 2
                        want = [ucol[min(max(k0 + (sp + 2) * Wt + 4 * w + t, k0), k1 - 1)] for t in range(4)]
                        assert [int(x) for x in ent[line, 0, 28:32]] == [int(x) for x in want]
                    for k in range(4 * ng):
                        slot, rsub = (int(meta[k]) >> 11) & 0x7f, (int(meta[k]) & 0xff) >> 2
                        if slot == 64:                                      # pad: -0.0 times the zero slice
                            assert vals[k] == np.int32(-2147483648) and rsub == 0
                            continue
                        assert slot < Wt and (rsub, slot) > prev            # by row, then column, strictly
                        prev = (rsub, slot)
                        r = int(rows[tile * 256 + rsub * 16 + w])
                        assert r >= 0
                        out.append((r, int(ucol[k0 + sp * Wt + slot]), int(vals[k])))
                    line += 1
                    first = False
                    if last:
                        break
    return out


@pytest.mark.parametrize('n,m,nnz,hubs,ratio,piece,block,seed', [
    (700, 900, 30000, 3, 0.9, 4, 52.0, 1), (300, 5000, 20000, 2, 0.8, 0, 8.0, 2), (1000, 400, 60000, 0, 1.0, 512, 200.0, 3),
    (257, 64, 3000, 1, 1.0, 2, 52.0, 4)])
def test_strip_plan_decodes_to_the_block(n, m, nnz, hubs, ratio, piece, block, seed, monkeypatch):
    import spmm_cpp
    A, (row, col, val) = _block(n, m, nnz, hubs, seed)
    plans = []
    for limit in (2 ** 31, 0):                                              # bitmap unions, then the sort path
        monkeypatch.setattr(spmm_cpp, '_STRIP_BITMAP_LIMIT', limit)
        S = spmm_cpp.StripPlan(A, ratio, piece, min_tile_nnz=32, block_target=block)
        assert S.npiece > 0
        got = _decode(S)
        rest = S.rest
        rrow = np.repeat(np.arange(n), np.diff(rest.rowptr.numpy()))
        got += list(zip(rrow.tolist(), rest.col.numpy().tolist(), rest.val.numpy().view(np.int32).tolist()))
        want = sorted(zip(row.tolist(), col.tolist(), val.view(np.int32).tolist()))
        assert sorted(got) == want                                          # every nonzero once, nothing else
        in_strips = S.strip_rows().numpy()
        assert not np.any(in_strips[rrow])                                  # a row is whole in one of the two
        assert abs(S.share - (1 - rest.nnz / A.nnz)) < 1e-12 and S.nu == int(S.ucol.numel())
        plans.append(S)
    a, b = plans
    for name in ('desc', 'rows', 'ucol', 'ent', 'ebase', 'cut'):
        assert torch.equal(getattr(a, name), getattr(b, name)), name
