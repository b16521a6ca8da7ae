/*
 * dgnn_spmm.h -- C ABI of libdgnn_spmm.so, the MI355X (gfx950) replacement for
 * dist-gnn's native SpMM operator.
 *
 * Every entry point takes plain device pointers and sizes, an explicit HIP
 * stream (passed as void* so the header needs no HIP include; NULL = the
 * null stream), and returns an int status: 0 = ok, nonzero = DGNN_E_* below.
 * After a nonzero return dgnn_last_error() gives a thread-local message.
 * Nothing here synchronises the host with the device and nothing allocates
 * device memory: all launches are asynchronous on `stream`.
 *
 * Reference interfaces replaced (paths relative to the reference repository):
 *   light_dist_gnn/spmm_cpp/spmm.cpp:17-33          spmm_cusparse (cusparseSpMM, COO_ALG4)
 *   light_dist_gnn/spmm_cpp/spmm_original.cpp:114-193 spmm_cusparse (coo2csr + csrmm2)
 *   1D_CAGNET/sparse-extension/sparse_coo_tensor.cpp:115-196 spmm_gpu (alpha=beta=1)
 *   light_dist_gnn/models/gcn.py:9-13               the `spmm` marshalling lambda
 */
#ifndef DGNN_SPMM_H
#define DGNN_SPMM_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DGNN_OK            0
#define DGNN_E_BADARG      1  /* null pointer, negative size, misaligned or aliasing operands */
#define DGNN_E_HIP         2  /* a HIP runtime call or a kernel launch failed                 */
#define DGNN_E_NODEVICE    3  /* no gfx950 device visible to this process                     */

/* ABI version of this header: a plain counter (250 now); the Python stub refuses a library of another one. */
int dgnn_version(void);

/* Message for the last nonzero status returned on the calling thread ("" if none). */
const char* dgnn_last_error(void);

/*
 * Sorted COO row indices -> CSR row pointer.
 *   row[nnz]     int32, non-decreasing, each in [0, n)   (what coalesce() guarantees,
 *                light_dist_gnn/coo_graph/graph_utils.py:48)
 *   rowptr[n+1]  int32 out: rowptr[i] = number of k with row[k] < i
 * Replaces cusparseXcoo2csr (spmm_original.cpp:141-146, sparse_coo_tensor.cpp:142-147),
 * which the reference re-runs (with a cudaMalloc) on every SpMM call.
 */
int dgnn_coo2csr_i32(const int32_t* row, int64_t nnz, int32_t n,
                     int32_t* rowptr, void* stream);

/*
 * C[n x F] = alpha * A[n x m] * B[m x F] + beta * C        (fp32, row-major, in place)
 *   A in CSR: rowptr[n+1], col[nnz], val[nnz]; columns of a row in ascending order
 *   B: ldb floats between rows (ldb >= F); C: ldc floats between rows (ldc >= F); columns F..ldc-1 of C
 *   are neither read nor written by any entry point
 *   beta == 0 never reads C (C may hold NaN/uninitialised memory).
 * B and C must not overlap. Each output element is accumulated in ascending
 * nonzero order by one lane, with fused multiply-adds: results are
 * run-to-run identical (no atomics).
 * Replaces cusparseSpMM / cusparseScsrmm2 (spmm.cpp:26-28, spmm_original.cpp:168-184).
 */
int dgnn_spmm_csr_f32(const int32_t* rowptr, const int32_t* col, const float* val,
                      int32_t n, int32_t m, int64_t nnz,
                      const float* B, int64_t ldb,
                      float* C, int64_t ldc, int32_t F,
                      float alpha, float beta, void* stream);

/*
 * Same operation on row-sorted COO input, the exact operand list of
 * spmm_cusparse (spmm.cpp:16-17): row[nnz], col[nnz], val[nnz].
 * `rowptr_ws` is caller-provided workspace of (n+1) int32 that receives the CSR
 * row pointer (coo2csr + csr spmm on the same stream).
 */
int dgnn_spmm_coo_f32(const int32_t* row, const int32_t* col, const float* val,
                      int32_t n, int32_t m, int64_t nnz,
                      int32_t* rowptr_ws,
                      const float* B, int64_t ldb,
                      float* C, int64_t ldc, int32_t F,
                      float alpha, float beta, void* stream);

/*
 * int64 COO indices (what torch.sparse_coo_tensor.indices() holds) -> int32.
 * Replaces the two `A.indices()[k].int()` copies the reference makes on every
 * call (light_dist_gnn/models/gcn.py:11). Values must fit int32; checked on the
 * host by the caller (sizes), not on the device.
 */
int dgnn_idx_i64_to_i32(const int64_t* src, int64_t count, int32_t* dst, void* stream);

/*
 * dst[r, 0:F] = src[r, 0:F], dst[r, F:ldd] = 0 for r < n. Produces the 16-byte-aligned feature
 * block (ldd % 4 == 0; ldd % 32 == 0 makes every row start on a 128-byte line) that
 * dgnn_spmm_plan_f32 gathers from. In the reference this slot is the `local_feature.clone()`
 * into the broadcast buffer (light_dist_gnn/models/gcn.py:22-23).
 */
int dgnn_pad_rows_f32(const float* src, int64_t lds, float* dst, int64_t ldd,
                      int32_t n, int32_t F, void* stream);

/*
 * dst[r, 0:F] = src[idx[r], 0:F], dst[r, F:ldd] = 0 for r < k: packs the rows of a feature block that
 * one peer's adjacency block references, already at the padded pitch dgnn_spmm_plan_f32 reads
 * (ldd % 4 == 0, dst 16-byte aligned). idx[k] int32, each in [0, nsrc) -- NOT checked on the device.
 * Replaces the advanced-indexing copy `t[needed_rows_idx]` of the needed-rows-only transfer
 * (1D_CAGNET/dist_1d.py:33-50 p2p_broadcast, same in dist_15d.py:35-52).
 */
int dgnn_gather_rows_f32(const float* src, int64_t lds, int32_t nsrc, const int32_t* idx, int32_t k,
                         float* dst, int64_t ldd, int32_t F, void* stream);

/*
 * Work plan of one CSR block for dgnn_spmm_plan_f32 (built once per block; the host-side
 * builder is dist-gnn_amd/spmm_cpp: SegPlan). Integer data only.
 *
 * The column range [0, m) is cut into `nstage` equal blocks. A segment is the run of one
 * row's nonzeros that falls into one column block: positions [k0, k1) of col/val. Runs
 * longer than the plan's piece length are cut into pieces; piece p of a cut run has
 * out = pbase + p (a row of the `partial` buffer), an uncut run has out = -1.
 *   seg            device, int32[4 * nseg]: {row, k0, k1, out} per entry, grouped by stage,
 *                  longest first inside a stage
 *   stage_seg_off  HOST, int64[nstage + 1]: entries of stage s are [off[s], off[s+1])
 *   cut            device, int32[4 * ncut]: {row, pbase, pieces, stride - 1} per cut run, grouped by stage: piece p of the
 *                  run is partial row pbase + p * stride (stride 1 here: 0; row bundles: rows_per_bundle)
 *   stage_cut_off  HOST, int64[nstage + 1]
 *   npartial       rows the `partial` buffer must hold (sum of pieces over all cut runs)
 *   covers_rows    1 if the FIRST stage lists every row the caller wants written (every row of the block; or every row
 *                  but those another call owns: spmm_cpp.SegPlan(skip_rows=...) beside dgnn_spmm_strip_overwrite_f32)
 *                  -- rows with nothing in the first
 *                  column block as empty runs (k0 == k1, out = -1) -- so that a product which overwrites C
 *                  (beta = 0, dgnn_spmm_plan_overwrite_f32) needs no zero fill: every row is stored by the
 *                  first stage kernel or by its merge. 0: such a product zero-fills C first.
 * No counterpart in the reference (cuSPARSE keeps its own internal buffers, spmm.cpp:26-28).
 */
typedef struct dgnn_plan {
  int32_t nstage;
  const int32_t* seg;
  const int64_t* stage_seg_off;
  const int32_t* cut;
  const int64_t* stage_cut_off;
  int64_t npartial;
  int32_t max_seg_len;      /* longest k1 - k0 in `seg` (> 0); bounds the critical path of a stage */
  int32_t covers_rows;
  /* Hot-tile plans (hot_rows > 0; 0 = none, the three pointers are then ignored). Per stage the plan names the
   * hot_rows most referenced columns of that column block; the kernel keeps their B rows (one feature chunk at a
   * time) in LDS. Such a plan indexes its OWN copy of the entry arrays -- pass them as `col` / `val` -- in which
   * every (row, stage) run lists its hot entries first, `col` holding the slot 0..hot_rows-1 instead of the column,
   * then its other entries with their columns; both parts in ascending column order.
   *   hot_rows   256 (feature chunks of 128 floats) or 512 (chunks of 64 floats)
   *   hot_cols   device, int32[nstage * hot_rows]: column of every slot (unused slots: any valid column)
   *   seg_hot    device, int32[nseg]: for segment i the position kh, k0 <= kh <= k1: entries [k0, kh) are hot
   * An output element then sums, inside a stage, its hot nonzeros first and the others after them (one chain). */
  int32_t hot_rows;
  const int32_t* hot_cols;
  const int32_t* seg_hot;
} dgnn_plan;

/*
 * A dgnn_plan built on the device, on the stream, from the CSR arrays alone (no host pass over the
 * nonzeros, no host synchronisation): column-block stages by binary search in each row's ascending
 * columns, runs grouped by length class (longest class first), long runs cut into pieces. The host never
 * learns the counts, so the lists are sized by upper bounds and unused slots are dead (segment row = -1,
 * cut count = 0; the kernels skip them). Workspace `ws` (int32, 16-byte aligned, ws_ints long):
 *   [0, hdr_ints)                                   counters
 *   then nstage lists of nseg_max segments (4 ints each), stage s at s * nseg_max
 *   then nstage lists of ncut_max cut runs  (4 ints each), stage s at s * ncut_max
 * so a dgnn_plan over it has seg = ws + hdr_ints, stage_seg_off[s] = s * nseg_max,
 * cut = ws + hdr_ints + 4 * nstage * nseg_max, stage_cut_off[s] = s * ncut_max, npartial = npartial_max,
 * max_seg_len = piece. nstage / piece = 0 ask for the library's rules.
 * The list order depends on the run (atomic slot reservation); results computed with it do not.
 * No counterpart in the reference.
 */
typedef struct dgnn_plan_layout {
  int32_t nstage, piece;
  int64_t hdr_ints, nseg_max, ncut_max, npartial_max, ws_ints;
} dgnn_plan_layout;
int dgnn_plan_layout_for(int32_t n, int32_t m, int64_t nnz, int64_t ldb, int32_t nstage, int32_t piece,
                         dgnn_plan_layout* out);
int dgnn_plan_build(const int32_t* rowptr, const int32_t* col, int32_t n, int32_t m,
                    const dgnn_plan_layout* layout, int32_t* ws, void* stream);

/*
 * C[n x F] = alpha * A * B + beta * C with a work plan: the fast path.
 *   col/val        the CSR arrays the plan indexes
 *   B              ldb >= F, ldb % 4 == 0, 16-byte aligned (see dgnn_pad_rows_f32); columns
 *                  F..ldb-1 are read but never stored
 *   partial        device scratch, npartial rows of ldp floats (ldp >= ldb, ldp % 4 == 0,
 *                  16-byte aligned); may be NULL when the plan cuts nothing
 * Launches, on `stream`: an optional beta pre-scale, then per stage one SpMM kernel and (if the
 * stage cut runs) one merge kernel. Same accumulation chain as dgnn_spmm_csr_f32 for every
 * uncut row; cut rows add their piece sums in piece order (deterministic).
 * Replaces cusparseSpMM (spmm.cpp:26-28) for callers that keep the block resident across
 * calls, i.e. the P-stage loop of light_dist_gnn/models/gcn.py:21-29.
 */
int dgnn_spmm_plan_f32(const dgnn_plan* plan, const int32_t* col, const float* val,
                       int32_t n, int32_t m, const float* B, int64_t ldb,
                       float* C, int64_t ldc, float* partial, int64_t ldp,
                       int32_t F, float alpha, float beta, void* stream);

/*
 * C = alpha * A * B for a C that need not be initialised (what the first stage of an aggregation wants: the
 * `z_loc = zeros` of gcn.py:18 and its read-back disappear). With a plan that covers every row in its first
 * stage nothing is zero-filled and C is never read: two passes over C less than zeros + accumulate, a tenth of
 * the traffic on graphs with a few nonzeros per row. Otherwise C is zero-filled first. Same bits as
 * dgnn_spmm_plan_f32(beta = 0). (dgnn_spmm_plan_f32 with beta = 0 takes the same route.)
 */
int dgnn_spmm_plan_overwrite_f32(const dgnn_plan* plan, const int32_t* col, const float* val,
                                 int32_t n, int32_t m, const float* B, int64_t ldb,
                                 float* C, int64_t ldc, float* partial, int64_t ldp,
                                 int32_t F, float alpha, void* stream);

/*
 * Row bundles: the fast path for blocks whose rows share many columns (hub columns; measured on the Reddit shape:
 * 0.74x the gathers, -13..-20 % time). The stage kernel gathers one slice of a B row per nonzero, and the unit that
 * serves gathers is what bounds it; here R rows (2, 4, 6 or 8; taken longest first inside a stage) are multiplied
 * together: the union of their columns is walked once and every gathered slice feeds up to R accumulators.
 *   desc            device, int32[(R + 4) * ndesc]: {row_0 .. row_(R-1) (-1 = none), k0, k1, out, 0} per bundle piece,
 *                   grouped by stage, longest first; entries [k0, k1) of ucol/uval; out < 0: the rows accumulate into
 *                   C, else the piece sums go to partial rows out + r and are merged per row (cut list, stride R)
 *   stage_desc_off  HOST, int64[nstage + 1]
 *   ucol            device, int32[nu]: union columns of a bundle, ascending
 *   uval            device, float[nu * R], 16-byte aligned: entry k's value for member r at uval[k * R + r], 0 where
 *                   that row has no nonzero in the column
 *   cut, stage_cut_off, npartial   as in dgnn_plan ({row, pbase, pieces, R - 1})
 * The first stage lists every row of the block. Every row sums exactly its own nonzeros in ascending column order
 * (a zero weight leaves the accumulator as it is for finite B), so rows of uncut bundles carry the bits of
 * dgnn_spmm_csr_f32; rows of cut bundles the fixed-order piece sums. No counterpart in the reference.
 */
typedef struct dgnn_bundle_plan {
  int32_t nstage;
  int32_t rows_per_bundle;
  const int32_t* desc;
  const int64_t* stage_desc_off;
  const int32_t* ucol;
  const float* uval;
  const int32_t* cut;
  const int64_t* stage_cut_off;
  int64_t npartial;
} dgnn_bundle_plan;
int dgnn_spmm_bundle_f32(const dgnn_bundle_plan* plan, int32_t n, int32_t m,
                         const float* B, int64_t ldb, float* C, int64_t ldc, float* partial, int64_t ldp,
                         int32_t F, float alpha, float beta, void* stream);

/*
 * Dense tiles: the part of a block that is not sparse. On graphs with hub rows and hub columns (Reddit shape: rows
 * taken 32 at a time in order of length, the columns at least 6 of the 32 reference: 47 % of the nonzeros at 39 %
 * density) a gather per nonzero is the wrong currency; there one coalesced slice of a B row feeds 32 rows' fused
 * multiply-adds whose A operands are wave-uniform (scalar loads). The host-side builder (spmm_cpp.DensePlan) moves
 * those entries out of the CSR block -- the sparse remainder goes through dgnn_spmm_plan_f32 / dgnn_spmm_bundle_f32
 * first -- and this call ADDS the dense part:  C += alpha * A_dense * B.
 *   rows_per_tile  32
 *   rows           device, int32[32 * ntile]: member rows of every tile (-1 = none)
 *   kcol           device, int32[nk]: the listed columns (slots), piece after piece, ascending inside a piece
 *   kval           device, float[32 * nk]: value of member r in slot k at kval[32 * k + r], 0 where the row has no such
 *                  nonzero
 *   desc           device, int32[4 * npiece]: {k0, k1, tile, out} per piece (slots [k0, k1) of one tile's list; k1 - k0 a
 *                  positive multiple of 8 -- pad slots repeat a column with zero weights; kcol/kval hold at least 8
 *                  slots), in working order; out < 0: the members accumulate into C, else the piece sums go to partial
 *                  rows out + r and are merged per row (cut list {row, first partial row, pieces, 31}, as in dgnn_plan)
 *   npartial       rows the partial buffer must hold
 * One wave takes one piece and one chunk of 128 floats of the feature width (64 floats when ldb <= 64).
 * An output element is one chain of fused multiply-adds over the tile's list in ascending column order, continuing
 * from what C holds (a zero weight leaves it as it is for finite B): rows of uncut tiles carry the bits of the strict
 * chain over [remainder entries, then listed entries]; rows of cut tiles the fixed-order piece sums. No MFMA, no
 * atomics. No counterpart in the reference.
 */
typedef struct dgnn_dense_plan {
  int32_t rows_per_tile;
  int32_t ntile;
  int32_t npiece;
  int32_t ncut;
  const int32_t* desc;
  const int32_t* rows;
  const int32_t* kcol;
  const float* kval;
  const int32_t* cut;
  int64_t nk;
  int64_t npartial;
} dgnn_dense_plan;
int dgnn_spmm_dense_f32(const dgnn_dense_plan* plan, int32_t n, int32_t m,
                        const float* B, int64_t ldb, float* C, int64_t ldc, float* partial, int64_t ldp,
                        int32_t F, float alpha, void* stream);

/*
 * Row strips: tiles of 256 rows (ranked by length, so that a tile's rows share the hub columns) whose union column list
 * is streamed ONCE through LDS in strips of 64 columns (LDS-DMA, double-buffered) -- "feature tiles of H_j staged in
 * LDS" (north_star). Each of the workgroup's 16 waves keeps 16 rows' accumulators for a 256-float feature chunk in
 * registers and walks a flat list of its (slot, row, value) entries per strip; the accumulator row is selected by VGPR
 * index mode, the B operand comes from LDS (two ds_read_b64 per entry and wave), so the address unit sees one 1-KiB slice
 * per (tile, column) instead of one per nonzero (Reddit shape: 0.15 of the nonzeros). The host-side builder
 * (spmm_cpp.StripPlan) takes the tiles whose union is short enough out of the CSR block -- the rest goes through
 * dgnn_spmm_plan_f32 first -- and this call ADDS the strip part:  C += A_strip * B  (alpha must be 1).
 *   rows_per_tile 256, strip_cols 64 (the widest strip), waves 16, line_entries 32 (the kernel's geometry; anything else
 *           is DGNN_E_BADARG)
 *   rows    device, int32[256 * ntile]: member i of tile t (-1 = none); wave w owns members w, w + 16, w + 32, ...
 *   ucol    device, int32[nu]: union columns, tile after tile, ascending inside a tile
 *   desc    device, int32[8 * npiece]: {tile, k0, k1, out, strips, columns per strip (8, 16, 32 or 64), 0, 0} per piece
 *           = consecutive strips of union entries [k0, k1) of one tile, in working order; out < 0: the members accumulate
 *           into C, else the piece sums go to partial rows out + i and are merged per row (cut list {row, first partial
 *           row, pieces, 255}, as in dgnn_plan)
 *   ebase   device, int64[16 * npiece]: first line of (piece, wave)'s entry stream
 *   ent     device, 256-byte aligned, uint32[64 * nline (+ 256 of slack)]: lines of 32 entries; dword k of a line is entry
 *           k's meta word (slot << 11 | member row / 16 * 4), dword 32 + k its value bits; the entries of (piece, wave,
 *           strip) are a block of whole lines, blocks in strip order; a line's first meta word also carries, in bits 20-23,
 *           the groups of 4 entries to take from the line (1..8) and, in bit 24, "last line of its block"; the FIRST line
 *           of the block of strip s holds at most 28 entries and, in meta words 28..31, the columns of the wave's slots
 *           4 w .. 4 w + 3 of strip s + 2 (clamped into [k0, k1 - 1]); pad entries {64 << 11, 0x80000000 (-0.0)}
 *   npartial  rows the partial buffer must hold
 * Every row sums its own nonzeros in ascending column order starting from what C holds, one fused multiply-add each:
 * rows of uncut tiles carry the bits of dgnn_spmm_csr_f32; rows of cut tiles the fixed-order piece sums. Pad entries
 * change nothing (fma(-0.0, +0.0, acc) == acc for every acc). No MFMA, no atomics. No counterpart in the reference
 * (spmm.cpp:26-28 is the operation).
 */
typedef struct dgnn_strip_plan {
  int32_t rows_per_tile;
  int32_t strip_cols;
  int32_t waves;
  int32_t line_entries;
  int32_t ntile;
  int32_t npiece;
  int32_t ncut;
  int32_t reserved;
  const int32_t* desc;
  const int32_t* rows;
  const int32_t* ucol;
  const uint32_t* ent;
  const int64_t* ebase;
  const int32_t* cut;
  int64_t nu;
  int64_t nline;
  int64_t npartial;
} dgnn_strip_plan;
int dgnn_spmm_strip_f32(const dgnn_strip_plan* plan, int32_t n, int32_t m,
                        const float* B, int64_t ldb, float* C, int64_t ldc, float* partial, int64_t ldp,
                        int32_t F, float alpha, void* stream);
/*
 * The same product for a C that holds nothing yet:  C[rows of the plan's tiles] = A_strip * B -- those rows are written
 * without being read (the tiles' rows are whole in the plan, so nothing else contributes to them); every other row of C is
 * left alone. The caller's plan for the remaining rows must then not list the tiles' rows (spmm_cpp.SegPlan(skip_rows=...)):
 * together the two calls write every row exactly once, where plan-of-the-rest (overwrite) + dgnn_spmm_strip_f32 write the
 * tiles' rows twice and read them once. Same bits (a chain that starts from +0.0 either way). Replaces z_loc =
 * torch.zeros(...) (gcn.py:18) plus the first stage for the rows of the tiles.
 */
int dgnn_spmm_strip_overwrite_f32(const dgnn_strip_plan* plan, int32_t n, int32_t m,
                        const float* B, int64_t ldb, float* C, int64_t ldc, float* partial, int64_t ldp,
                        int32_t F, float alpha, void* stream);

/*
 * spmm_cusparse's operand list (row-sorted COO, spmm.cpp:16-17) on the fast kernel, for callers that hand
 * over a fresh COO every call as the reference's models do (gcn.py:11) and so cannot keep a plan: a
 * work plan is built on the stream, on the device (column-block stages; per stage the runs grouped by
 * length class, longest first; runs longer than the piece length cut and merged as in
 * dgnn_spmm_plan_f32). No host synchronisation: the
 * lists and the partial buffer are sized by upper bounds that dgnn_coo_aligned_workspace reports.
 * Uncut rows carry the bits of dgnn_spmm_coo_f32; cut rows the fixed-order piece sums.
 *   B           ldb % 4 == 0, 16-byte aligned base (dgnn_pad_rows_f32)
 *   rowptr_ws   int32[n + 1] workspace
 *   plan_ws     int32[plan_ws_ints] workspace, 16-byte aligned
 *   partial     float[partial_rows * ldp] workspace, ldp >= ldb, ldp % 4 == 0, 16-byte aligned
 */
int dgnn_coo_aligned_workspace(int32_t n, int32_t m, int64_t nnz, int64_t ldb,
                               int64_t* plan_ws_ints, int64_t* partial_rows);
int dgnn_spmm_coo_aligned_f32(const int32_t* row, const int32_t* col, const float* val,
                              int32_t n, int32_t m, int64_t nnz, int32_t* rowptr_ws, int32_t* plan_ws,
                              const float* B, int64_t ldb, float* C, int64_t ldc,
                              float* partial, int64_t ldp, int32_t F,
                              float alpha, float beta, void* stream);

/*
 * dgnn_spmm_coo_aligned_f32 for callers that cannot allocate or carry workspaces: code under HIP graph
 * capture, tight serving loops, harnesses that forbid hipMalloc. NO COUNTERPART IN THE REFERENCE. Same
 * operands and results as dgnn_spmm_coo_aligned_f32 (same kernels), minus the three workspace arguments:
 * row pointer, device-built work plan and partial rows live in a file-scope device array of
 * DGNN_NOALLOC_SCRATCH_BYTES that every GPU the library runs on holds until the process ends. The call
 * allocates and frees nothing and does not synchronise.
 *   limit     dgnn_noalloc_scratch_needed(n, m, nnz, ldb) <= DGNN_NOALLOC_SCRATCH_BYTES, else DGNN_E_BADARG
 *             (roughly: up to ~1 M rows, any nnz < 2^31, at 256-float rows; dgnn_noalloc_limits reports the size)
 *   sharing   all calls on one GPU use that GPU's one array: calls that overlap in time on one GPU (two
 *             streams, two threads, a graph replay next to an eager call) RACE. Serialise them.
 *   B         ldb >= F, ldb % 4 == 0, 16-byte aligned base, as for dgnn_spmm_coo_aligned_f32
 *   set-up    the first call on a device looks the array's address up (hipGetSymbolAddress: NOT a stream operation,
 *             it may load the code object). It must happen OUTSIDE a graph capture: call dgnn_noalloc_limits() -- which
 *             does the same look-up for the current device -- or one eager dgnn_spmm_coo_noalloc_f32 on every device
 *             first. After that a call only enqueues kernels on `stream` and can be captured and replayed
 *             (tests/test_zzz_no_alloc.py::test_noalloc_entry_point_under_graph_capture).
 *   cost      the array is part of the code object: EVERY process that loads the library holds 64 MiB on every GPU it
 *             touches, whether or not it ever calls this entry point.
 */
#define DGNN_NOALLOC_SCRATCH_BYTES (64u << 20)
int dgnn_noalloc_limits(int64_t* scratch_bytes);
int dgnn_noalloc_scratch_needed(int32_t n, int32_t m, int64_t nnz, int64_t ldb, int64_t* bytes);
int dgnn_spmm_coo_noalloc_f32(const int32_t* row, const int32_t* col, const float* val,
                              int32_t n, int32_t m, int64_t nnz,
                              const float* B, int64_t ldb, float* C, int64_t ldc, int32_t F,
                              float alpha, float beta, void* stream);

/*
 * Measurement hooks (no counterpart in the reference). Between dgnn_prof_begin() and dgnn_prof_end...() every kernel
 * launch of the library is bracketed by HIP events on its own stream. dgnn_prof_end() waits for them and returns the
 * summed durations of the SpMM kernels (stage_*) and of the helper kernels -- pad, gather, beta scale, merge --
 * (other_*); dgnn_prof_end_by_kernel() returns them per kernel: ms[k], launches[k] for k < count, k one of DGNN_PROF_*.
 * Thread-safe (the record list is mutex-guarded); adds two event records per launch while enabled.
 */
#define DGNN_PROF_STAGE 0   /* spmm_stage_kernel */
#define DGNN_PROF_BUNDLE 1  /* spmm_bundle_kernel */
#define DGNN_PROF_DENSE 2   /* spmm_dense_kernel */
#define DGNN_PROF_STRIP 3   /* spmm_strip_kernel */
#define DGNN_PROF_HOT 4     /* spmm_hot_kernel */
#define DGNN_PROF_ROWS 5    /* spmm_rows_kernel */
#define DGNN_PROF_MERGE 6   /* merge_kernel (partial rows of cut runs / bundles / tiles) */
#define DGNN_PROF_HELPER 7  /* pad, gather, scale */
#define DGNN_PROF_COUNT 8
int dgnn_prof_begin(void);
int dgnn_prof_end(double* stage_ms, int64_t* stage_launches, double* other_ms, int64_t* other_launches);
int dgnn_prof_end_by_kernel(double* ms, int64_t* launches, int32_t count);

#ifdef __cplusplus
}
#endif
#endif /* DGNN_SPMM_H */
