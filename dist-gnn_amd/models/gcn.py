"""Distributed GCN layer over a 1D row partition (CAGNET 1D).

Drop-in for light_dist_gnn/models/gcn.py: ``spmm`` (:9-13), ``broadcast`` (:16-30),
``DistGCNLayer`` (:33-53), ``GCN`` (:56-72) keep their names, arguments and results.

What changed underneath:
  * ``spmm(A, B, C)`` runs the hand-written gfx950 kernels through the C ABI (spmm_cpp). Each
    adjacency block is converted to int32 CSR + a work plan once and cached on the block,
    instead of two int64->int32 index copies per call (gcn.py:11). There is NO fallback to
    ``C.addmm_(A, B)`` (gcn.py:13): a missing library or a CPU tensor raises.
  * ``broadcast`` is a pipeline: the stage-(j+1) feature block travels over RCCL while the
    stage-j SpMM runs (``inflight`` receive buffers rotate; RCCL runs on its own stream and is
    ordered against the compute stream by events, never by device-wide synchronisation).
    Stages are still accumulated in the fixed order src = 0..P-1, so results do not depend on
    timing and equal the reference's sequential loop.
"""
import os

import torch
import torch.distributed as dist
import torch.nn as nn
import torch.nn.functional as F

from dist_utils import DistEnv, gemm as _gemm

_spmm_override = None          # tests install the CPU oracle here to exercise the pipeline on gloo
_gather_override = None


def set_spmm_backend(fn, gather=None):
    """Replace the SpMM used by the pipeline (``fn(A_sparse_coo, B, C)`` doing C += A @ B in place) and,
    optionally, the row gather of the needed-rows transfer (``gather(src, idx_int32, dst)``).
    ``None`` restores the HIP operators. Test hook: the product never installs a fallback itself."""
    global _spmm_override, _gather_override
    _spmm_override = fn
    _gather_override = gather


def _trace(*a):
    if os.environ.get('DGNN_TRACE') == '1':
        import sys
        import time
        print(f'[dgnn {time.perf_counter():.3f}]', *a, file=sys.stderr, flush=True)


def _gather_rows(src, idx, dst):
    """dst[r, :F] = src[idx[r]] (dst may be wider: zero fill) with the HIP kernel dgnn_gather_rows_f32."""
    if _gather_override is not None:
        return _gather_override(src, idx, dst)
    import spmm_cpp
    return spmm_cpp.gather_rows(src, idx, dst)


class NeededRows:
    """Needed-rows-only transfer (1D_CAGNET/dist_1d.py:33-50 ``p2p_broadcast`` with
    ``nz_col_dict``, dist_data.py:96-101): block A[rank, src] only reads the rows of H_src whose column
    holds a nonzero, so only those travel. Built once per block list:

      cols[src]      sorted distinct columns of A[rank, src]          (what this rank needs from src)
      blocks[src]    A[rank, src] with its columns renumbered 0..k-1 in the same order: the received
                     rows ARE its B operand, nothing is scattered back. The renumbering is monotone,
                     so every row keeps its nonzeros in the same order and z stays bit-identical.
      send_idx[dst]  rows of this rank's H that dst needs (int32), learnt in one all-to-all at set-up
      K[dst][src]    row counts, known to every rank (sizes of every message)
    """

    def __init__(self, adj_parts, group, local=None):
        env = DistEnv.env
        P, rank = env.world_size, env.rank
        if dist.get_world_size(group) != P or len(adj_parts) != P:
            raise RuntimeError('NeededRows: only the 1D layout over the world group is supported')
        dev = adj_parts[0].device
        self.cols, self.blocks = local if local is not None else self._local(adj_parts, rank)
        self._exchange(adj_parts, group, P, rank, dev)

    @staticmethod
    def _local(adj_parts, rank):
        """The part of the set-up that needs no peer (and can fail on one rank alone): the distinct columns and the
        renumbered block of every source."""
        dev = adj_parts[0].device
        cols_of, blocks = [], []
        for src in range(len(adj_parts)):
            A = adj_parts[src] if adj_parts[src].is_coalesced() else adj_parts[src].coalesce()
            idx = A.indices()
            if src == rank:                       # the own block reads the local features in place
                cols_of.append(torch.zeros(0, dtype=torch.long, device=dev))
                blocks.append(A)
                continue
            cols, inv = torch.unique(idx[1], sorted=True, return_inverse=True)
            blk = torch.sparse_coo_tensor(torch.stack([idx[0], inv]), A.values(), (A.size(0), max(int(cols.numel()), 1)),
                                          is_coalesced=True)   # still sorted by (row, column), no duplicates
            cols_of.append(cols)
            blocks.append(blk)
        return cols_of, blocks

    def _exchange(self, adj_parts, group, P, rank, dev):
        """Row counts (all-gather) and index lists (one all-to-all): collectives, every rank together."""
        counts = torch.tensor([c.numel() for c in self.cols], dtype=torch.long, device=dev)
        every = [torch.empty_like(counts) for _ in range(P)]
        dist.all_gather(every, counts, group=group)
        self.K = torch.stack(every).cpu()                    # K[dst][src]
        want = torch.cat([c.to(torch.int32) for c in self.cols]) if P > 1 else torch.zeros(0, dtype=torch.int32, device=dev)
        in_split = [int(x) for x in self.K[rank]]
        out_split = [int(x) for x in self.K[:, rank]]
        got = torch.empty(sum(out_split), dtype=torch.int32, device=dev)
        dist.all_to_all_single(got, want, output_split_sizes=out_split, input_split_sizes=in_split, group=group)
        self.send_idx = list(got.split(out_split))
        self.send_idx_all = got                             # the same lists back to back, in rank order
        self.block_rows = adj_parts[0].size(1)
        self.rank, self._compact = rank, None
        # the same number on every rank (K is): ranks must agree on whether this form is worth trying
        self.fraction = float(self.K.sum()) / max(1, P * (P - 1) * self.block_rows)
        self.local_fraction = float(self.K[rank].sum()) / max(1, (P - 1) * self.block_rows)

    def compact_slab(self):
        """The rank's whole row slab as ONE matrix over the compact operand
        [needed rows of H_0 | ... | all rows of the own block | ... | needed rows of H_(P-1)] (source order), built on
        first use: one SpMM instead of P, z written once instead of read and written per source. Columns are
        renumbered monotonically inside each block and the blocks keep their source order, so every row keeps its
        nonzeros in the order of the P block products -- same chain, same bits. Returns (matrix, row offset of every
        source in the operand, operand rows, send index list with the own block's rows in place)."""
        if self._compact is None:
            P, rank = len(self.blocks), self.rank
            dev = self.blocks[0].device
            offs, at = [], 0
            for src in range(P):
                offs.append(at)
                at += self.block_rows if src == rank else int(self.cols[src].numel())
            rows = torch.cat([b.indices()[0] for b in self.blocks])
            cols = torch.cat([b.indices()[1] + offs[src] for src, b in enumerate(self.blocks)])
            vals = torch.cat([b.values() for b in self.blocks])
            order = torch.argsort(rows * max(at, 1) + cols)
            slab = torch.sparse_coo_tensor(torch.stack([rows[order], cols[order]]), vals[order],
                                           (self.blocks[0].size(0), max(at, 1)), is_coalesced=True)
            own = torch.arange(self.block_rows, dtype=torch.int32, device=dev)
            send_idx = torch.cat(list(self.send_idx[:rank]) + [own] + list(self.send_idx[rank + 1:]))
            self._compact = (slab, offs, at, send_idx)
        return self._compact

    @staticmethod
    def fraction_needed(adj_parts, group):
        """Share of the feature rows that have to travel at all, over the whole job (same value on every
        rank: one all-reduce). Cheap test before building anything: one bincount per block."""
        got = getattr(adj_parts, '_dgnn_needed_fraction', None)
        if got is not None:
            return got
        env = DistEnv.env
        need = 0
        for src, A in enumerate(adj_parts):
            if src != env.rank and A._nnz():
                A = A if A.is_coalesced() else A.coalesce()
                need += int(torch.count_nonzero(torch.bincount(A.indices()[1], minlength=A.size(1))))
        dev = adj_parts[0].device if dist.get_backend(group) == 'nccl' else 'cpu'
        t = torch.tensor([float(need)], dtype=torch.float64, device=dev)
        dist.all_reduce(t, group=group)
        P = env.world_size
        got = float(t.item()) / max(1, P * (P - 1) * adj_parts[0].size(1))
        try:
            adj_parts._dgnn_needed_fraction = got
        except AttributeError:
            pass
        return got

    @staticmethod
    def local_part(adj_parts, group):
        """Build (and keep) this rank's own half of the set-up: no collective, so its failure can be voted on."""
        if getattr(adj_parts, '_dgnn_needed', None) is None and getattr(adj_parts, '_dgnn_needed_local', None) is None:
            local = NeededRows._local(adj_parts, DistEnv.env.rank)
            try:
                adj_parts._dgnn_needed_local = local
            except AttributeError:
                pass

    @staticmethod
    def of(adj_parts, group):
        nr = getattr(adj_parts, '_dgnn_needed', None)
        if nr is None:
            nr = NeededRows(adj_parts, group, local=getattr(adj_parts, '_dgnn_needed_local', None))
            if hasattr(adj_parts, '_dgnn_needed_local'):
                del adj_parts._dgnn_needed_local
            try:
                adj_parts._dgnn_needed = nr
            except AttributeError:                # a plain list: rebuilt per call (tests only)
                pass
        return nr


def _csr_block(A):
    """CsrBlock cached on the sparse tensor object (built on first use)."""
    import spmm_cpp
    blk = getattr(A, '_dgnn_csr', None)
    if blk is None:
        if not A.is_cuda:
            raise RuntimeError('spmm: adjacency block is on the CPU; the HIP SpMM has no CPU path '
                               '(models.gcn.set_spmm_backend is the test hook for gloo runs)')
        blk = spmm_cpp.CsrBlock.from_coo(A if A.is_coalesced() else A.coalesce())
        A._dgnn_csr = blk
    return blk


def spmm(A, B, C, fresh=False):
    """C += A @ B in place (alpha = beta = 1, as gcn.py:11 calls spmm_cusparse). ``fresh``: nothing has been
    added to C yet (the first stage of an aggregation) -- C = A @ B, whatever C holds: the HIP path neither
    zero-fills nor reads it (see _new_output); a test backend gets a zeroed C and adds."""
    if _spmm_override is not None:
        if fresh:
            C.zero_()
        return _spmm_override(A, B, C)
    import spmm_cpp
    spmm_cpp.aggregate(_csr_block(A), B, C, overwrite=fresh)


class _Sum:
    """z = sum of stage products, in the order given: the first product that is actually issued overwrites z
    (no zero fill, no read-back), the others accumulate; a sum with no term at all is zero."""

    def __init__(self, z):
        self.z, self.started = z, False

    def add(self, A, B):
        spmm(A, B, self.z, fresh=not self.started)
        self.started = True

    def done(self):
        if not self.started:
            self.z.zero_()
        return self.z


def _row_padded(t):
    import spmm_cpp
    return spmm_cpp.is_row_padded(t)


def _new_output(rows, feat, padded):
    """The [rows, F] output of an aggregation. On the HIP path it is NOT zero-filled: the first stage product
    overwrites it (_Sum). ``padded``: a [rows, F] view of a buffer at the kernel's 128-byte row pitch (aligned
    float4 read-modify-write in the SpMM, aligned leading dimension for the GEMM that consumes it); columns past
    F are never read or written. DGNN_POISON=1 (tests) fills it with NaN so that any element left unwritten shows."""
    F = feat.size(1)
    hip = feat.is_cuda and _spmm_override is None
    ld = (F + 31) // 32 * 32 if (padded and hip) else F
    buf = (torch.empty if hip else torch.zeros)((rows, ld), dtype=feat.dtype, device=feat.device)
    if hip and os.environ.get('DGNN_POISON') == '1':
        buf.fill_(float('nan'))
    return buf[:, :F] if ld != F else buf


class StagePipeline:
    """The P-stage broadcast + SpMM loop of gcn.py:16-30, overlapped.

    mode 'bcast'      P rooted broadcasts, ``inflight`` (>= 2) receive buffers: while stage j is
                      multiplied, stages j+1 .. j+inflight-1 are already in flight.
    mode 'allgather'  one all-gather of every rank's block (the P broadcasts of a layer pass
                      are together an all-gather; one collective keeps every xGMI link busy),
                      then ONE SpMM of the rank's whole row slab against the gathered matrix when the
                      block list carries the slab (Parted_COO_Graph sets it; DGNN_FUSED_SLAB=0 or a
                      plain list fall back to the P per-block SpMMs).
    mode 'agpipe'     the all-gather cut into two column slabs: slab 2 travels while slab 1 is multiplied
                      (collective bandwidth and overlap; needs at least 64 feature columns).
                      'agpipe4': four slabs (more overlap, narrower SpMMs; a candidate from 256 columns up).
    mode 'needed'     point-to-point: the owner of stage j sends every peer only the rows of H_j
                      that peer's block references (NeededRows), pipelined like 'bcast'.
    mode 'needed_a2a' the same rows in ONE all-to-all (all links at once), then the P compact-block SpMMs;
                      'needed_a2a2': in two column slabs, slab 2 travelling while slab 1 is multiplied.
    mode 'auto'       (default) the first time a (tag, shape) is aggregated, the forms are run
                      and timed on the spot ('needed' joins when under 3/4 of the rows are
                      needed), the slowest rank's time decides (all-reduce MAX, so every rank
                      picks the same form), and the choice sticks. Costs a few extra aggregations
                      in the first epoch; results do not depend on the choice.
    All accumulate in ascending source (= column) order, so every row the work plans leave uncut
    gets the same bits in every mode.

    Wire format: when padding a feature row to the kernel's 128-byte row pitch adds <= 1/16 to the
    bytes (602 -> 608 floats), the owner pads its block once and the padded block travels, so the
    receivers' SpMMs read it in place; otherwise the [s, F] block travels as is and each receiver
    pads its copy (dgnn_pad_rows_f32).

    ``cache_input`` (opt-in, DGNN_CACHE_INPUT=1; the reference's plain GCN re-sends every epoch and only its
    CachedGCN reuses blocks): the layer-1 forward pass moves the
    *input features*, which do not change between epochs. When the same tensor OBJECT, unwritten since (the
    entry holds a reference to it and its version counter), is aggregated again under the same tag, the blocks received last time
    are reused and nothing is sent -- lossless, results identical, one of the 2L transfers of an
    epoch removed. (The reference's plain GCN re-sends them every epoch; its CachedGCN does this
    by policy, cached_gcn.py:26.)
    """

    def __init__(self, mode=None, inflight=None, cache_input=None):
        self.mode = mode or os.environ.get('DGNN_PIPELINE', 'auto')
        if self.mode not in ('auto', 'bcast', 'allgather', 'agpipe', 'agpipe4', 'needed', 'needed_a2a', 'needed_a2a2'):
            raise ValueError(f'DGNN_PIPELINE={self.mode!r}: expected auto, bcast, allgather, agpipe, agpipe4, needed, '
                             'needed_a2a or needed_a2a2')
        self.inflight = max(1, int(inflight or os.environ.get('DGNN_INFLIGHT', 2)))   # 1 = no overlap (debugging)
        self.cache_input = (os.environ.get('DGNN_CACHE_INPUT', '0') == '1') if cache_input is None else cache_input
        self._bufs = {}
        self._input_cache = {}         # tag -> (key of the source tensor, [block per src])
        self.transfers = 0             # blocks actually moved (tests / logs)
        self.rows_moved = 0            # feature rows received by this rank ...
        self.rows_full = 0             # ... and what whole blocks would have been
        self.fused_slab = os.environ.get('DGNN_FUSED_SLAB', '1') == '1'
        self.choice = {}               # (tag, rows, width, group size) -> (form, {form: seconds}) picked by 'auto'
        self.first_use = {}            # same key -> {form: seconds of its first run, set-up excluded} (logs, bench line)

    def _buffers(self, shape, like, count, tag=''):
        key = (tuple(shape), like.dtype, like.device, tag)
        have = self._bufs.get(key, [])
        while len(have) < count:
            have.append(torch.empty(shape, dtype=like.dtype, device=like.device))
        self._bufs[key] = have
        return have

    @staticmethod
    def _same_input(hit, t):
        """The cache entry holds the source tensor itself (so its storage cannot be freed and handed to another
        tensor while the entry lives) and its version counter at the time: a hit needs the same object, unwritten."""
        return hit[0] is t and hit[1] == t._version

    @staticmethod
    def _wire_width(feat):
        """Row pitch (floats) of the block on the wire: the kernel's padded pitch when that is cheap."""
        F = feat.size(1)
        if not feat.is_cuda or _spmm_override is not None or feat.dtype != torch.float32:
            return F
        ld = (F + 31) // 32 * 32
        return ld if (ld - F) * 16 <= F else F

    def _own_block(self, feat, ld):
        """This rank's block as it travels (contiguous [s, ld]): feat itself, its rows with their padding when it is
        stored at the padded pitch already (spmm_cpp.row_padded: the input features), or a padded copy."""
        s, F = feat.shape
        if ld == F:
            return feat.contiguous()
        if (feat.stride(1) == 1 and feat.stride(0) == ld and feat.data_ptr() % 128 == 0
                and (feat.storage_offset() + s * ld) * feat.element_size() <= feat.untyped_storage().nbytes()):
            return feat.as_strided((s, ld), (ld, 1))      # the columns past F travel too; they only ever meet the
        import spmm_cpp                                    # padding columns of a product
        return spmm_cpp.pad_rows(feat, self._buffers((s, ld), feat, 1, 'own')[0])

    def describe(self):
        """What 'auto' settled on, for logs and the bench line."""
        if self.mode != 'auto':
            return self.mode
        if not self.choice:
            return 'auto'
        return 'auto(' + ', '.join(f'{k[0]}[{k[1]}x{k[2]}]={v[0]}' for k, v in sorted(self.choice.items())) + ')'

    # ---- the forms -------------------------------------------------------------------
    def _run_allgather(self, adj_parts, feat, stages, group, z, keep):
        env = DistEnv.env
        rank, timer = env.rank, env.timer
        P, s, F = len(stages), feat.size(0), feat.size(1)
        ld = self._wire_width(feat)
        own = self._own_block(feat, ld)
        gathered = (self._buffers((P * s, ld), feat, 1, 'gather')[0] if keep is None
                    else feat.new_empty((P * s, ld)))          # kept blocks must not be overwritten
        with timer.timing_cuda('broadcast'):
            dist.all_gather_into_tensor(gathered, own, group=group)
        self.transfers += P                                  # same count as P rooted broadcasts
        slab = getattr(adj_parts, 'slab', None) if self.fused_slab else None
        if slab is not None and slab.size(1) == P * s and [src for src, _ in stages] == list(range(P)):
            # the gathered matrix is H in global row order and the rank's row slab A[rank, :] has global
            # column ids: ONE SpMM instead of P -- z is written once instead of read and written per
            # stage (what dominates on sparse graphs), same ascending-column chain for every row
            if keep is not None:
                keep['gathered'] = gathered
            with timer.timing_cuda('spmm'):
                spmm(slab, gathered[:, :F] if ld != F else gathered, z, fresh=True)
            return z
        members = sorted(src for src, _ in stages)
        acc = _Sum(z)
        for src, blk in stages:
            pos = members.index(src)
            block = gathered[pos * s:(pos + 1) * s, :F]
            if keep is not None and src != rank:
                keep[src] = block
            with timer.timing_cuda('spmm'):
                acc.add(adj_parts[blk], block)
        return acc.done()

    def _run_agpipe(self, adj_parts, feat, stages, group, z, chunks=2):
        """All-gather in column chunks: the feature columns are cut into `chunks` slabs (at multiples of 32
        floats) and the all-gather of slab c + 1 travels while slab c is multiplied -- the collective keeps
        every xGMI link busy AND overlaps the SpMM. Columns are independent, so the bits do not change."""
        timer = DistEnv.env.timer
        P, s, F = len(stages), feat.size(0), feat.size(1)
        step = (-(-F // chunks) + 31) // 32 * 32                  # columns per slab: ceil(F / chunks), on a 32-float grid
        bounds = [(c0, min(c0 + step, F)) for c0 in range(0, F, step)]
        if len(bounds) < 2:
            return self._run_allgather(adj_parts, feat, stages, group, z, None)
        on_gpu = feat.is_cuda and _spmm_override is None
        slab = getattr(adj_parts, 'slab', None) if self.fused_slab else None
        fused = slab is not None and slab.size(1) == P * s and [src for src, _ in stages] == list(range(P))
        members = sorted(src for src, _ in stages)
        pending = {}

        def post(c):
            c0, c1 = bounds[c]
            w = c1 - c0
            ld = (w + 31) // 32 * 32 if on_gpu else w
            if on_gpu and (ld != w or c0 % 4):
                import spmm_cpp
                own = spmm_cpp.pad_rows(feat[:, c0:c1], self._buffers((s, ld), feat, 1, f'agp-own{c}')[0])
            else:
                own = self._buffers((s, ld), feat, 1, f'agp-own{c}')[0]
                own.copy_(feat[:, c0:c1])
            gathered = self._buffers((P * s, ld), feat, 1, f'agp-gather{c}')[0]
            pending[c] = (dist.all_gather_into_tensor(gathered, own, group=group, async_op=True), gathered, w)

        post(0)
        for c, (c0, c1) in enumerate(bounds):
            if c + 1 < len(bounds):
                post(c + 1)
            work, gathered, w = pending.pop(c)
            with timer.timing_cuda('broadcast'):
                work.wait()
            zc = z[:, c0:c1]
            with timer.timing_cuda('spmm'):
                if fused:
                    spmm(slab, gathered[:, :w], zc, fresh=True)
                else:
                    acc = _Sum(zc)
                    for src, blk in stages:
                        pos = members.index(src)
                        acc.add(adj_parts[blk], gathered[pos * s:(pos + 1) * s, :w])
                    acc.done()
        self.transfers += P
        return z

    def _run_bcast(self, adj_parts, feat, stages, group, z, keep, policy):
        env = DistEnv.env
        rank, timer = env.rank, env.timer
        cs = torch.cuda.current_stream(feat.device) if feat.is_cuda else None
        n, F = len(stages), feat.size(1)
        ld = self._wire_width(feat)
        own = None
        shape = (feat.size(0), ld)
        depth = min(self.inflight, max(n, 1))
        # a kept block must not be overwritten by a later stage: give every stage its own buffer then
        bufs = (self._buffers(shape, feat, depth, 'recv') if keep is None
                else [torch.empty(shape, dtype=feat.dtype, device=feat.device) for _ in range(n)])
        pending = {}

        def post(i):
            """Start moving the block of stage i: returns at once, the transfer runs on RCCL's stream."""
            nonlocal own
            src = stages[i][0]
            if policy is not None:
                hit = policy.cached(src)
                if hit is not None:
                    pending[i] = (None, hit)
                    return
            if src == rank:
                if own is None:
                    own = self._own_block(feat, ld)
                buf = own
            else:
                buf = bufs[i % len(bufs)]
            work = dist.broadcast(buf, src=src, group=group, async_op=True)
            self.transfers += 1
            pending[i] = (work, buf[:, :F] if ld != F else buf)

        acc = _Sum(z)
        for i in range(min(depth - 1, n)):
            post(i)
        for i, (src, blk) in enumerate(stages):
            if i + depth - 1 < n:
                post(i + depth - 1)
            work, block = pending.pop(i)
            _trace('bcast wait', rank, 'stage', i, 'src', src)
            with timer.timing_cuda('broadcast', cs):
                if work is not None:
                    work.wait()
            _trace('bcast got', rank, 'stage', i)
            if policy is not None and work is not None:
                policy.arrived(src, block)
            if keep is not None and src != rank:
                keep[src] = block
            with timer.timing_cuda('spmm', cs):
                acc.add(adj_parts[blk], block)
        return acc.done()

    def _run_needed(self, adj_parts, feat, stages, group, z):
        env = DistEnv.env
        rank, timer = env.rank, env.timer
        nr = NeededRows.of(adj_parts, group)
        n, F = len(stages), feat.size(1)
        if feat.is_cuda and _gather_override is None:
            ld = self._wire_width(feat)
            ld = ld if ld != F else (F + 3) // 4 * 4          # the gather kernel writes float4s
        else:
            ld = F
        K = nr.K
        depth = min(self.inflight, max(n, 1))
        kmax = max(1, int(K[rank].max()))
        recv = self._buffers((kmax, ld), feat, depth, 'needed-recv')
        out_rows = [int(K[dst][rank]) for dst in range(len(K))]
        send = self._buffers((max(1, sum(out_rows)), ld), feat, 1, 'needed-send')[0]
        pending, sends = {}, []

        def post(i):
            src = stages[i][0]
            if src == rank:
                ops, at = [], 0
                for dst, k in enumerate(out_rows):
                    if dst == rank or k == 0:
                        continue
                    part = send[at:at + k]
                    at += k
                    _gather_rows(feat, nr.send_idx[dst], part)
                    ops.append(dist.P2POp(dist.isend, part, dst, group))
                sends.extend(dist.batch_isend_irecv(ops) if ops else [])
                pending[i] = ([], None)
                return
            k = int(K[rank][src])
            if k == 0:
                pending[i] = ([], None)
                return
            buf = recv[i % len(recv)][:k]
            pending[i] = (dist.batch_isend_irecv([dist.P2POp(dist.irecv, buf, src, group)]), buf)
            self.rows_moved += k

        acc = _Sum(z)
        for i in range(min(depth - 1, n)):
            post(i)
        for i, (src, blk) in enumerate(stages):
            if i + depth - 1 < n:
                post(i + depth - 1)
            works, buf = pending.pop(i)
            with timer.timing_cuda('broadcast'):
                for w in works:
                    w.wait()
            self.transfers += 1
            if src == rank:
                with timer.timing_cuda('spmm'):
                    acc.add(adj_parts[blk], feat)
                continue
            self.rows_full += nr.block_rows
            if buf is not None:
                with timer.timing_cuda('spmm'):
                    acc.add(nr.blocks[src], buf[:, :F] if ld != F else buf)
        for w in sends:                       # the send buffer is reused by the next aggregation
            w.wait()
        return acc.done()

    def _run_needed_a2a(self, adj_parts, feat, stages, group, z, chunks=1):
        """The needed rows of every peer in ONE all-to-all (every xGMI link busy at once; the staged
        point-to-point form above keeps only the stage owner's links busy), then the P block SpMMs on the
        compact blocks. ``chunks`` = 2 cuts the feature columns into two slabs (at a multiple of 32 floats):
        slab 2 travels while slab 1 is multiplied. Same accumulation order as every other form."""
        env = DistEnv.env
        rank, timer = env.rank, env.timer
        nr = NeededRows.of(adj_parts, group)
        P, F = len(stages), feat.size(1)
        K = nr.K
        in_rows = [int(K[rank][src]) for src in range(P)]        # what arrives from src
        out_rows = [int(K[dst][rank]) for dst in range(P)]       # what goes to dst
        on_gpu = feat.is_cuda and _gather_override is None
        step = (-(-F // chunks) + 31) // 32 * 32 if chunks > 1 else F
        bounds = [(c0, min(c0 + step, F)) for c0 in range(0, F, step)]
        pending = {}
        # Fused (default, DGNN_FUSED_SLAB=0 turns it off): the own block travels through the all-to-all as this
        # rank's message to itself (a device-local copy), so the receive buffer IS the compact operand in source
        # order and ONE SpMM of the compact row slab replaces the P block products (see NeededRows.compact_slab).
        fused = self.fused_slab and [src for src, _ in stages] == list(range(P))
        send_idx = nr.send_idx_all
        if fused:
            slab, _, _, send_idx = nr.compact_slab()
            in_rows[rank] = out_rows[rank] = nr.block_rows

        def post(c):
            c0, c1 = bounds[c]
            w = c1 - c0
            ld = (w + 31) // 32 * 32 if on_gpu else w
            send = self._buffers((max(1, sum(out_rows)), ld), feat, 1, f'a2a-send{c}')[0][:sum(out_rows)]
            recv = self._buffers((max(1, sum(in_rows)), ld), feat, 1, f'a2a-recv{c}')[0][:sum(in_rows)]
            if sum(out_rows):
                _gather_rows(feat[:, c0:c1] if w != F else feat, send_idx, send)
            work = dist.all_to_all_single(recv, send, output_split_sizes=in_rows, input_split_sizes=out_rows,
                                          group=group, async_op=True)
            pending[c] = (work, recv, w, ld)

        post(0)
        for c, (c0, c1) in enumerate(bounds):
            if c + 1 < len(bounds):
                post(c + 1)
            work, recv, w, ld = pending.pop(c)
            with timer.timing_cuda('broadcast'):
                work.wait()
            if fused:
                with timer.timing_cuda('spmm'):
                    spmm(slab, recv[:, :w] if ld != w else recv, z[:, c0:c1] if w != F else z, fresh=True)
                continue
            acc = _Sum(z[:, c0:c1] if w != F else z)
            at = 0
            with timer.timing_cuda('spmm'):
                for src, blk in stages:
                    if src == rank:
                        acc.add(adj_parts[blk], feat[:, c0:c1] if w != F else feat)
                        continue
                    k = in_rows[src]
                    if k:
                        rows = recv[at:at + k]
                        acc.add(nr.blocks[src], rows[:, :w] if ld != w else rows)
                        at += k
                acc.done()
        self.transfers += P
        self.rows_moved += sum(in_rows) - (nr.block_rows if fused else 0)
        self.rows_full += nr.block_rows * (P - 1)
        return z

    def _prepare(self, form, adj_parts, feat, stages, group):
        """Everything of a form that can fail on one rank alone: receive / gather / send buffers, the CSR blocks and
        work plans of the matrices it multiplies at the widths it multiplies them, the compact row slab. NO collective
        is posted here, so a rank that raises can still tell its peers in the vote that follows (_pick_form)."""
        env = DistEnv.env
        P, s, F = len(stages), feat.size(0), feat.size(1)
        on_gpu = feat.is_cuda and _spmm_override is None
        slab = getattr(adj_parts, 'slab', None) if self.fused_slab else None
        fused = slab is not None and slab.size(1) == P * s and [src for src, _ in stages] == list(range(P))

        def plans(mats, widths):
            if on_gpu:
                import spmm_cpp
                for A in mats:
                    for w in widths:
                        spmm_cpp._plan_for(_csr_block(A), w)

        def slabs(chunks):
            step = (-(-F // chunks) + 31) // 32 * 32 if chunks > 1 else F
            return [min(c0 + step, F) - c0 for c0 in range(0, F, step)]

        if form == 'bcast':
            self._buffers((s, self._wire_width(feat)), feat, min(self.inflight, max(P, 1)), 'recv')
            plans([adj_parts[blk] for _, blk in stages], [F])
        elif form == 'allgather':
            self._buffers((P * s, self._wire_width(feat)), feat, 1, 'gather')
            plans([slab] if fused else [adj_parts[blk] for _, blk in stages], [F])
        elif form in ('agpipe', 'agpipe4'):
            ws = slabs(4 if form == 'agpipe4' else 2)
            for c, w in enumerate(ws):
                ld = (w + 31) // 32 * 32 if on_gpu else w
                self._buffers((s, ld), feat, 1, f'agp-own{c}')
                self._buffers((P * s, ld), feat, 1, f'agp-gather{c}')
            plans([slab] if fused else [adj_parts[blk] for _, blk in stages], sorted(set(ws)))
        else:                                               # needed, needed_a2a, needed_a2a2
            nr = NeededRows.of(adj_parts, group)            # built (with its set-up exchange) before the trials
            ws = slabs(2 if form == 'needed_a2a2' else 1)
            if form == 'needed':
                plans([nr.blocks[src] if src != env.rank else adj_parts[blk] for src, blk in stages], [F])
            elif self.fused_slab and [src for src, _ in stages] == list(range(P)):
                plans([nr.compact_slab()[0]], sorted(set(ws)))
            else:
                plans([nr.blocks[src] if src != env.rank else adj_parts[blk] for src, blk in stages], sorted(set(ws)))

    def _pick_form(self, adj_parts, feat, stages, group, tag):
        """'auto': run and time every candidate form once for this (tag, shape) and keep the fastest.

        What can go wrong with a form, and what happens then:
          * its set-up raises on some rank (memory for a gather buffer, a plan)  -> _prepare runs WITHOUT collectives
            and is followed by a MAX all-reduce vote: every rank drops the form, the job goes on;
          * it raises after _prepare, i.e. possibly after posting a collective its peers are now waiting in -> fatal
            (the communicator is out of step; a vote could itself hang): supervise.give_up names the form and the
            rank leaves; under the supervising launcher (bench.py, dist_utils/supervise.py) fresh rank processes
            start without the form, otherwise the exception is re-raised;
          * it hangs -> the per-trial deadline (DGNN_FORM_DEADLINE_S, armed by the launcher) ends the rank the same way;
          * it is slower than DGNN_AUTO_CAP_S on first use -> dropped by the vote.
        DGNN_AUTO_EXCLUDE=form[,form] removes candidates ('bcast', the reference's own form and the fallback of all
        others, cannot be excluded)."""
        key = (tag, feat.size(0), feat.size(1), len(stages))
        hit = self.choice.get(key)
        if hit is not None:
            return hit[0]
        import time
        from dist_utils import supervise
        sent = self.transfers
        log = DistEnv.env.logger.log
        # the votes travel as host tensors under gloo (its CUDA path stages through the host anyway)
        vote_dev = feat.device if dist.get_backend(group) == 'nccl' else 'cpu'

        def vote(*xs):
            v = torch.tensor(xs, dtype=torch.float64, device=vote_dev)
            dist.all_reduce(v, op=dist.ReduceOp.MAX, group=group)
            return [float(x) for x in v.tolist()]

        forms = ['bcast']
        if self._gather_fits(feat, len(stages)):            # the gathered matrix is P times a block: papers-sized
            forms += ['allgather'] + (['agpipe'] if feat.size(1) > 32 else [])   # graphs cannot afford it
            forms += ['agpipe4'] if feat.size(1) >= 256 else []                   # four slabs of >= 64 columns
        if self._needed_ok(adj_parts, stages, group) and NeededRows.fraction_needed(adj_parts, group) < 0.75:
            # The staged point-to-point form (batch_isend_irecv per stage) has only ever run over gloo; over RCCL
            # it is a candidate only on request (DGNN_AUTO_P2P=1). The all-to-all forms are plain collectives.
            p2p = dist.get_backend(group) != 'nccl' or os.environ.get('DGNN_AUTO_P2P', '0') == '1'
            forms += (['needed'] if p2p else []) + ['needed_a2a'] + (['needed_a2a2'] if feat.size(1) >= 64 else [])
        out = supervise.excluded()
        if out:
            forms = [f for f in forms if f == 'bcast' or f not in out]
        if any(f.startswith('needed') for f in forms):
            # the index lists are exchanged once, by all ranks together, and only if every rank built its own
            # (NeededRows.local_part: no collective) -- vote first, exchange after
            try:
                NeededRows.local_part(adj_parts, group)
                bad = 0.0
            except Exception as e:                          # noqa: BLE001
                log(f'pipeline auto {tag}: needed-rows set-up raised {type(e).__name__}: {e} -- forms dropped', rank=-1)
                bad = 1.0
            if vote(bad)[0]:
                forms = [f for f in forms if not f.startswith('needed')]
            else:
                with supervise.deadline('needed_a2a'):
                    NeededRows.of(adj_parts, group)
        moved = (self.rows_moved, self.rows_full)

        def once(form):
            z = _new_output(adj_parts[stages[0][1]].size(0), feat, False)
            if feat.is_cuda:
                torch.cuda.synchronize(feat.device)
            t0 = time.perf_counter()
            if form == 'bcast':
                self._run_bcast(adj_parts, feat, stages, group, z, None, None)
            elif form == 'allgather':
                self._run_allgather(adj_parts, feat, stages, group, z, None)
            elif form in ('agpipe', 'agpipe4'):
                self._run_agpipe(adj_parts, feat, stages, group, z, chunks=4 if form == 'agpipe4' else 2)
            elif form in ('needed_a2a', 'needed_a2a2'):
                self._run_needed_a2a(adj_parts, feat, stages, group, z, chunks=2 if form == 'needed_a2a2' else 1)
            else:
                self._run_needed(adj_parts, feat, stages, group, z)
            if feat.is_cuda:
                torch.cuda.synchronize(feat.device)
            return time.perf_counter() - t0

        def trial(form, first=False):
            """One run of a form on every rank -> the slowest rank's seconds; None when its set-up raised on any rank
            (first use only; every rank learns it from one MAX all-reduce, so all drop the same forms)."""
            with supervise.deadline(form):
                if first:
                    try:
                        self._prepare(form, adj_parts, feat, stages, group)
                        bad = 0.0
                    except Exception as e:                  # noqa: BLE001 -- whatever it is, the form is out
                        if form == 'bcast':                 # the reference's own form, everybody's fallback: the job's failure
                            raise
                        log(f'pipeline auto {tag}: set-up of form {form} raised {type(e).__name__}: {e} -- dropped', rank=-1)
                        bad = 1.0
                    if vote(bad)[0]:
                        return None
                try:
                    t = once(form)
                except Exception as e:                      # noqa: BLE001
                    log(f'pipeline auto {tag}: form {form} raised {type(e).__name__}: {e} after its set-up -- the ranks '
                        'may be out of step inside a collective: fatal', rank=-1)
                    if form != 'bcast' and os.environ.get('DGNN_SUPERVISED') == '1':
                        supervise.give_up(form, f'raised {type(e).__name__} after its set-up')
                    raise
                return vote(t)[0]

        cap = float(os.environ.get('DGNN_AUTO_CAP_S', 30.0))
        times, first_use = {}, {}
        with torch.no_grad():
            # line the ranks up first: the first epoch is also where each rank times its library GEMM candidates
            # (dist_utils.tune_gemms), seconds that differ from rank to rank and would otherwise be charged to the
            # first form tried (and could push it over the cap)
            vote(0.0)
            if feat.is_cuda:
                torch.cuda.synchronize(feat.device)
            alive = []
            for f in forms:                                 # builds CSR blocks, plans and buffers: not timed
                t = trial(f, first=True)
                if t is not None:
                    first_use[f] = t
                if t is not None and (f == 'bcast' or t <= cap):
                    alive.append(f)
                elif t is not None:
                    log(f'pipeline auto {tag}: form {f} took {t:.1f} s on first use (cap {cap:.0f} s) -- dropped', rank=0)
            for f in alive:
                # best of two timed runs per form: one sample is at the mercy of whatever else the node is doing
                times[f] = min(trial(f), trial(f))
        forms = [f for f in forms if f in times]
        form = min(forms, key=lambda f: (times[f], forms.index(f)))
        self.choice[key] = (form, times)
        self.first_use[key] = first_use
        self.rows_moved, self.rows_full = moved
        self.transfers = sent                               # tuning traffic is not part of the count
        DistEnv.env.logger.log(f'pipeline auto {tag} [{feat.size(0)}x{feat.size(1)}] P={len(stages)}: '
                               + ', '.join(f'{f} {times[f] * 1e3:.3f} ms' for f in forms) + f' -> {form}'
                               + (f' (excluded: {",".join(sorted(out))})' if out else ''), rank=0)
        return form

    @staticmethod
    def _gather_fits(feat, P):
        """The all-gather forms hold H of the whole graph on every rank (per feature width): only while that is
        at most 1/16 of the device memory (18 GB on an MI355X; a papers100M layer would need 53-106 GiB)."""
        if not feat.is_cuda:
            return True
        need = P * feat.size(0) * ((feat.size(1) + 31) // 32 * 32) * feat.element_size()
        return need * 16 <= torch.cuda.get_device_properties(feat.device).total_memory

    @staticmethod
    def _needed_ok(adj_parts, stages, group):
        env = DistEnv.env
        return (len(stages) == env.world_size > 1 and dist.get_world_size(group) == env.world_size
                and all(src == blk for src, blk in stages) and len(adj_parts) == env.world_size)

    def run(self, adj_parts, feat, tag, policy=None, group=None, stages=None, padded_out=False):
        """z = sum over stages of A[stage.block] @ H_(stage.src).

        ``padded_out``: z may be a [s, F] view of a wider zero-padded buffer (for callers that only feed
        it to matrix products, like DistGCNLayer); by default z is contiguous as in gcn.py:18.

        ``stages``: list of (src_rank, block_index) pairs, default [(j, j) for j in range(P)] -- the
        1D loop; the 1.5D variant passes the subset its replica owns and its column group.
        ``policy``: object with ``cached(src) -> tensor | None`` (asked before a transfer) and
        ``arrived(src, block)`` (told after one), used by the cached / adaptive variants.
        """
        env = DistEnv.env
        rank, timer = env.rank, env.timer
        group = env.world_group if group is None else group
        if stages is None:
            stages = [(j, j) for j in range(env.world_size)]
        given = feat
        # a matrix stored at the kernels' row pitch (the input features, spmm_cpp.row_padded) is read as it lies
        orig = feat if (feat.is_cuda and _spmm_override is None and _row_padded(feat)) else feat.contiguous()
        # Collectives run on a view with its own version counter: a rooted broadcast "writes" its buffer on
        # the root too as far as autograd can tell (gloo's CUDA path really does), and this tensor is often
        # saved for backward (a ReLU output). Its bits are never changed here.
        feat = orig.data
        rows = adj_parts[stages[0][1]].size(0) if stages else feat.size(0)
        z = _new_output(rows, feat, padded_out)
        plain = policy is None
        if plain and len(stages) == 1 and stages[0][0] == rank:
            with timer.timing_cuda('spmm'):
                spmm(adj_parts[stages[0][1]], feat, z, fresh=True)
            return z

        # a temporary (the contiguous copy of a strided input) is gone after this call: never cached
        cacheable = (plain and self.cache_input and tag.startswith('Forward') and tag.endswith('L1')
                     and not orig.requires_grad and orig is given)
        if cacheable:
            hit = self._input_cache.get(tag)
            if hit is not None and self._same_input(hit, orig):
                if 'gathered' in hit[2]:
                    with timer.timing_cuda('spmm'):
                        spmm(adj_parts.slab, hit[2]['gathered'][:, :feat.size(1)], z, fresh=True)
                    return z
                acc = _Sum(z)
                for src, blk in stages:
                    with timer.timing_cuda('spmm'):
                        acc.add(adj_parts[blk], feat if src == rank else hit[2][src])
                return acc.done()
        keep = {} if cacheable else None

        _trace('run', rank, tag, tuple(feat.shape), self.mode)
        form = self.mode
        can_gather = plain and len(stages) == dist.get_world_size(group) and len(stages) > 1
        if form == 'auto':
            form = self._pick_form(adj_parts, feat, stages, group, tag) if can_gather else 'bcast'
        if form == 'needed' and plain and self._needed_ok(adj_parts, stages, group):
            self._run_needed(adj_parts, feat, stages, group, z)
            keep = None                                     # compact blocks are not kept: nothing to reuse
        elif form in ('needed_a2a', 'needed_a2a2') and plain and self._needed_ok(adj_parts, stages, group):
            self._run_needed_a2a(adj_parts, feat, stages, group, z, chunks=2 if form == 'needed_a2a2' else 1)
            keep = None
        elif form in ('agpipe', 'agpipe4') and can_gather:
            self._run_agpipe(adj_parts, feat, stages, group, z, chunks=4 if form == 'agpipe4' else 2)
            keep = None
        elif form == 'allgather' and can_gather:
            self._run_allgather(adj_parts, feat, stages, group, z, keep)
        else:
            self._run_bcast(adj_parts, feat, stages, group, z, keep, policy)
        if keep is not None:
            self._input_cache[tag] = (orig, orig._version, keep)
        return z


_pipeline = None


def pipeline():
    global _pipeline
    if _pipeline is None:
        _pipeline = StagePipeline()
    return _pipeline


def grad_weight_product(features, ag):
    """features^T @ ag, the [F_in, F_out] weight gradient (gcn.py:50). With hundreds of thousands of rows this
    is a reduction over a very long K that the BLAS heuristics serve badly (1.59 ms for 233 K x 602 x 256
    on MI355X); cutting K into 4 batches for one bmm and adding the four partial products takes 0.69 ms.
    Same fp32 sums in a different association (8e-6 relative, measured); small blocks keep the single call."""
    n = features.size(0)
    if n < 65536 or not features.is_cuda:
        return _gemm.mm(features.t(), ag)
    k = 4
    m = n // k * k
    out = _gemm.bmm(features[:m].reshape(k, m // k, -1).transpose(1, 2), ag[:m].reshape(k, m // k, -1)).sum(0)
    if m < n:
        out = out + _gemm.mm(features[m:].t(), ag[m:])
    return out


def transform_first():
    """DGNN_TRANSFORM_FIRST=1: layers whose output is narrower than their input multiply by W before
    aggregating. Read at call time so tests can flip it."""
    return os.environ.get('DGNN_TRANSFORM_FIRST', '0') == '1'


def wgrad_from_z():
    """DGNN_WGRAD_FROM_Z=1 (opt-in): a layer whose INPUT needs no gradient (layer 1: the dataset's features) keeps its
    forward aggregation Z = A_hat H and takes the weight gradient as Z^T G -- the exact derivative of out = Z W --
    instead of aggregating the output gradient first, H^T (A_hat G) (gcn.py:47-50). One SpMM and one feature-block
    exchange less per epoch. The two agree when A_hat is symmetric (the reference's own assumption, gcn.py:47), in a
    different fp32 association (<= 1e-5 relative, tested); on an asymmetric graph this is the TRUE gradient while the
    reference's formula is not, so it is off by default (the compat default reproduces the reference there too)."""
    return os.environ.get('DGNN_WGRAD_FROM_Z', '0') == '1'


def broadcast(local_adj_parts, local_feature, tag, padded_out=False):
    """z_loc = sum over src of A[rank, src] @ H_src  (new [s, F] tensor), gcn.py:16-30."""
    return pipeline().run(local_adj_parts, local_feature, tag, padded_out=padded_out)


class DistGCNLayer(torch.autograd.Function):
    """out = (A_hat H) W with A_hat H aggregated across ranks; gcn.py:33-53.
    Backward reuses the same (un-transposed) blocks exactly like gcn.py:47 -- correct because the
    normalised adjacency of an undirected graph is symmetric. If the block list carries
    ``transposed`` blocks (Parted_COO_Graph(..., transposed_backward=True)) the backward pass
    aggregates with (A^T)[rank, :] instead, which is the true gradient on asymmetric graphs too."""

    @staticmethod
    def forward(ctx, features, weight, adj_parts, tag):
        ctx.save_for_backward(features, weight)
        ctx.adj_parts, ctx.tag = adj_parts, tag
        timer = DistEnv.env.timer
        if transform_first() and weight.size(1) < features.size(1):
            # opt-in A_hat (H W): aggregate at the narrower output width (Reddit layer 1: 256 instead of
            # 602 floats per gathered row). Same mathematics, different fp32 association than gcn.py:39-41,
            # so it is off by default; the backward formulas below hold unchanged. The tag suffix keeps
            # this temporary out of the input-feature cache.
            with timer.timing_cuda('mm'):
                hw = _gemm.mm(features, weight)
            return broadcast(adj_parts, hw, 'Forward' + tag + 'T')
        z_local = broadcast(adj_parts, features, 'Forward' + tag, padded_out=True)
        if wgrad_from_z() and not ctx.needs_input_grad[0]:
            ctx.z_local = z_local                       # kept instead of re-deriving A_hat G in the backward pass
        with timer.timing_cuda('mm'):
            return _gemm.mm(z_local, weight)

    @staticmethod
    def backward(ctx, grad_output):
        features, weight = ctx.saved_tensors
        z_local = getattr(ctx, 'z_local', None)
        if z_local is not None:                         # opt-in: dW = Z^T G, no aggregation of the output gradient
            ctx.z_local = None
            with DistEnv.env.timer.timing_cuda('mm'):
                grad_weight = grad_weight_product(z_local, grad_output)
            with DistEnv.env.timer.timing_cuda('all_reduce'):
                DistEnv.env.all_reduce_sum(grad_weight)
            return None, grad_weight, None, None
        blocks = getattr(ctx.adj_parts, 'transposed', None) or ctx.adj_parts
        ag = broadcast(blocks, grad_output, 'Backward' + ctx.tag, padded_out=True)
        with DistEnv.env.timer.timing_cuda('mm'):
            # the input features of layer 1 need no gradient: skip that product (the reference computes
            # and discards it, gcn.py:49); every gradient that is used is unchanged
            grad_features = _gemm.mm(ag, weight.t()) if ctx.needs_input_grad[0] else None
            grad_weight = grad_weight_product(features, ag)
        with DistEnv.env.timer.timing_cuda('all_reduce'):
            DistEnv.env.all_reduce_sum(grad_weight)
        return grad_features, grad_weight, None, None


def aggregation_layer(name, aggregate, forward_tag=lambda t: 'Forward' + t, backward_tag=lambda t: 'Backward' + t,
                      reduce_grad=None):
    """The layer of gcn.py:33-53 around another aggregation: ``out = aggregate(adj_parts, H, tag) @ W`` with the
    same backward formulas (aggregate grad_output with the same blocks, ``ag W^T`` only if the input needs it,
    ``H^T ag`` summed over the ranks by ``reduce_grad``, default the world all-reduce). The cached, adaptive and
    1.5D variants are this with their own ``aggregate``."""

    class Layer(torch.autograd.Function):
        @staticmethod
        def forward(ctx, features, weight, adj_parts, tag):
            ctx.save_for_backward(features, weight)
            ctx.adj_parts, ctx.tag = adj_parts, tag
            z_local = aggregate(adj_parts, features, forward_tag(tag))
            with DistEnv.env.timer.timing_cuda('mm'):
                return _gemm.mm(z_local, weight)

        @staticmethod
        def backward(ctx, grad_output):
            features, weight = ctx.saved_tensors
            ag = aggregate(ctx.adj_parts, grad_output, backward_tag(ctx.tag))
            with DistEnv.env.timer.timing_cuda('mm'):
                grad_features = _gemm.mm(ag, weight.t()) if ctx.needs_input_grad[0] else None
                grad_weight = grad_weight_product(features, ag)
            with DistEnv.env.timer.timing_cuda('all_reduce'):
                (reduce_grad or DistEnv.env.all_reduce_sum)(grad_weight)
            return grad_features, grad_weight, None, None

    Layer.__name__ = Layer.__qualname__ = name
    return Layer


class GCN(nn.Module):
    """ReLU GCN; ``num_layers=3`` is the reference model (gcn.py:56-72): weights drawn with
    torch.rand after torch.manual_seed(0) on the CPU, in layer order, then moved to the device."""

    layer_cls = DistGCNLayer

    def __init__(self, g, env, hidden_dim=16, num_layers=3):
        super().__init__()
        self.g, self.env, self.num_layers = g, env, num_layers
        dims = [g.features.size(1)] + [hidden_dim] * (num_layers - 1) + [g.num_classes]
        torch.manual_seed(0)
        for i in range(num_layers):
            setattr(self, f'weight{i + 1}', nn.Parameter(torch.rand(dims[i], dims[i + 1]).to(env.device)))
        if g.features.is_cuda and g.features.dtype == torch.float32 and _spmm_override is None:
            # the input features are aggregated every epoch and never change: store them at the kernels' row pitch
            # once (same values, a [n, F] view of a zero-padded buffer) instead of copying them there per epoch
            import spmm_cpp
            g.features = spmm_cpp.row_padded(g.features)

    def forward(self, features):
        h = features
        for i in range(self.num_layers):
            h = self.layer_cls.apply(h, getattr(self, f'weight{i + 1}'), self.g.adj_parts, f'L{i + 1}')
            if i + 1 < self.num_layers:
                h = F.relu(h)
        return h
