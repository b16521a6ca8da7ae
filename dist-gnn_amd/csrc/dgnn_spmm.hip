// dgnn_spmm.hip -- hand-written gfx950 (MI355X / CDNA4) kernels behind the C ABI in
// include/dgnn_spmm.h. Replaces dist-gnn's cuSPARSE wrapper
// (reference: light_dist_gnn/spmm_cpp/spmm.cpp:17-33, spmm_original.cpp:114-193).
//
// The SpMM  C = alpha*A*B + beta*C  is a gather: every nonzero (i, j, a) pulls the F floats of
// B[j, :] through the cache hierarchy and fuses them into the accumulators of C[i, :]. There is
// no MFMA here. What bounds it on MI355X is the vector-memory gather rate (L1/TA -> per-XCD L2
// -> Infinity Cache -> HBM); the kernels below are organised around keeping gathers L2-resident
// and many of them in flight -- and, where the block allows it, around paying fewer gathers per nonzero (row bundles,
// dense tiles). Kernel families:
//
//   spmm_rows_kernel   general operands (any alignment / leading dimension). One group of L
//                      lanes owns one output row; strictly ascending accumulation.
//   spmm_stage_kernel  the fast path: needs a work plan (dgnn_plan) and a B whose rows are
//                      16-byte aligned (ldb % 4 == 0). Feature chunks of 64 or 128 floats,
//                      XCD-aware chunk-major work order with balanced XCD shares, column-block
//                      stages, long rows cut in pieces, 16- or 32-lane groups.
//   spmm_bundle_kernel the same walk over the union of R rows' columns: one gather feeds up to R accumulators
//                      (blocks whose rows share columns).
//   spmm_dense_kernel  tiles of 32 rows x the columns several of them share, stored densely: one coalesced slice of a
//                      B row feeds 32 rows' packed FMAs whose A operands come through the scalar data path (hub rows x
//                      hub columns of power-law graphs); adds to what the sparse remainder of the block has written.
//
// In both, a (col, val) run of a row is read with coalesced loads, one entry per lane. The row
// kernel hands entries lane-to-lane (v_readlane when L == 64, ds_bpermute otherwise); the
// stage kernel parks the whole run as (32-bit B-row byte offset, alpha*val) pairs in a
// group-private LDS strip and reads it back two entries per ds_read_b128. Lane li owns VW
// consecutive floats of the feature chunk; every output element has exactly one accumulator
// that starts at beta*C and sees its nonzeros in ascending order with fused multiply-adds --
// the same chain as ATen's CPU sparse addmm (oracle/spmm_ref.c), so results are bit-identical
// to the oracle except for rows the plan cuts into pieces (fixed-order piece sums:
// deterministic, within fp32 rounding of the chain). No atomics anywhere.
//
// A work plan comes either from the host (spmm_cpp.SegPlan: exact longest-first order) or from the
// plan_*_kernel trio below, which builds the same structure on the device without the host ever
// learning a count (dgnn_plan_build; used by the plan-less reference call dgnn_spmm_coo_aligned_f32).
//
// 64-wide wavefronts are assumed throughout (gfx950 only).

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include <atomic>
#include <mutex>
#include <vector>

#include "dgnn_spmm.h"

namespace {

thread_local char g_err[512] = "";
// A/B knobs of tools/*_bench.py (lane-group widths, gathers in flight, parts of a kernel switched off for timing): they
// exist only in a library built with -DDGNN_EXPERIMENTS (lib/libdgnn_spmm_exp.so, declared in
// include/dgnn_spmm_experiments.h). In the product build the knob is a constant 0 and every branch on it folds away.
#ifdef DGNN_EXPERIMENTS
int g_variant = 0;
#else
constexpr int g_variant = 0;
#endif

int fail(int code, const char* fmt, const char* a = "", long long b = 0) {
  snprintf(g_err, sizeof(g_err), fmt, a, b);
  return code;
}

#define DGNN_HIP(call)                                                          \
  do {                                                                          \
    hipError_t e_ = (call);                                                     \
    if (e_ != hipSuccess)                                                       \
      return fail(DGNN_E_HIP, "%s (line %lld)", hipGetErrorString(e_), __LINE__); \
  } while (0)

// Optional in-library timing of the SpMM kernels with HIP events on the launch stream
// (dgnn_prof_begin / dgnn_prof_end): bench.py uses it for the live roofline figure.
// kinds: the SpMM kernels one by one, then the helpers (DGNN_PROF_* in the header)
enum { K_STAGE = 0, K_BUNDLE = 1, K_DENSE = 2, K_STRIP = 3, K_HOT = 4, K_ROWS = 5, K_MERGE = 6, K_HELPER = 7, K_COUNT = 8 };
struct ProfRec { hipEvent_t a, b; int kind; };
std::atomic<bool> g_prof_on{false};
std::mutex g_prof_mu;                         // launches may come from several host threads (forward / autograd engine)
std::vector<ProfRec> g_prof;

struct ProfScope {
  hipStream_t s; int kind; hipEvent_t a = nullptr, b = nullptr;
  ProfScope(hipStream_t s_, int kind_) : s(s_), kind(kind_) {
    if (g_prof_on.load(std::memory_order_relaxed) && hipEventCreate(&a) == hipSuccess && hipEventCreate(&b) == hipSuccess) (void)hipEventRecord(a, s);
    else a = nullptr;
  }
  ~ProfScope() {
    if (a) { (void)hipEventRecord(b, s); std::lock_guard<std::mutex> lk(g_prof_mu); g_prof.push_back({a, b, kind}); }
  }
};

template <int VW> struct Vec;
template <> struct Vec<1> { using T = float; };
template <> struct Vec<2> { using T = float2; };
template <> struct Vec<4> { using T = float4; };

template <int VW>
__device__ __forceinline__ void ld_vec(const float* p, float (&d)[VW]) {
  typename Vec<VW>::T t = *reinterpret_cast<const typename Vec<VW>::T*>(p);
  if constexpr (VW == 1) { d[0] = t; }
  if constexpr (VW == 2) { d[0] = t.x; d[1] = t.y; }
  if constexpr (VW == 4) { d[0] = t.x; d[1] = t.y; d[2] = t.z; d[3] = t.w; }
}

template <int VW>
__device__ __forceinline__ void st_vec(float* p, const float (&d)[VW]) {
  typename Vec<VW>::T t;
  if constexpr (VW == 1) { t = d[0]; }
  if constexpr (VW == 2) { t.x = d[0]; t.y = d[1]; }
  if constexpr (VW == 4) { t.x = d[0]; t.y = d[1]; t.z = d[2]; t.w = d[3]; }
  *reinterpret_cast<typename Vec<VW>::T*>(p) = t;
}

// Hand entry j of the group's staged (col, val) segment to every lane of the group.
template <int L>
__device__ __forceinline__ void bcast_entry(int cv, float vv, int g, int j, int& c, float& v) {
  if constexpr (L == 64) {
    c = __builtin_amdgcn_readlane(cv, j);                       // wave-uniform -> SGPR
    v = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(vv), j));
  } else {
    const int src = (g * L + j) << 2;
    c = __builtin_amdgcn_ds_bpermute(src, cv);
    v = __int_as_float(__builtin_amdgcn_ds_bpermute(src, __float_as_int(vv)));
  }
}

// ---------------------------------------------------------------------------------------
// General path: row-per-group CSR SpMM.
//   grid.x = ceil(n / (4 waves * G rows)), grid.y = feature chunks of L*VW*NPASS floats.
// ---------------------------------------------------------------------------------------
template <int VW, int L, int NPASS>
__global__ __launch_bounds__(256) void spmm_rows_kernel(
    const int32_t* __restrict__ rowptr, const int32_t* __restrict__ col,
    const float* __restrict__ val, int32_t n,
    const float* __restrict__ B, int64_t ldb,
    float* __restrict__ C, int64_t ldc, int32_t F, float alpha, float beta) {
  constexpr int G = 64 / L;          // rows per wave
  constexpr int U = 4;               // nonzeros whose gathers are in flight together
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int g = lane / L;
  const int li = lane % L;
  const int64_t row = ((int64_t)blockIdx.x * 4 + wave) * G + g;
  const int fbase = (blockIdx.y * NPASS * L + li) * VW;   // pass p covers fbase + p*L*VW

  int kb = 0, ke = 0;
  if (row < n) { kb = rowptr[row]; ke = rowptr[row + 1]; }

  bool fok[NPASS];
#pragma unroll
  for (int p = 0; p < NPASS; ++p) fok[p] = (fbase + p * L * VW) < F;

  float acc[NPASS][VW];
  float* const cr = C + row * ldc + fbase;
#pragma unroll
  for (int p = 0; p < NPASS; ++p) {
#pragma unroll
    for (int e = 0; e < VW; ++e) acc[p][e] = 0.f;
    if (beta != 0.f && row < n && fok[p]) {
      ld_vec<VW>(cr + p * L * VW, acc[p]);
      if (beta != 1.f) {
#pragma unroll
        for (int e = 0; e < VW; ++e) acc[p][e] *= beta;
      }
    }
  }

  const float* Bf = B + fbase;

  for (int k0 = kb; k0 < ke; k0 += L) {
    const int kk = k0 + li;
    int cv = 0; float vv = 0.f;
    if (kk < ke) { cv = col[kk]; vv = alpha * val[kk]; }
    const int cnt = min(L, ke - k0);
    int j = 0;
    for (; j + U <= cnt; j += U) {
      int c[U]; float v[U];
      float b[U][NPASS][VW];
#pragma unroll
      for (int u = 0; u < U; ++u) bcast_entry<L>(cv, vv, g, j + u, c[u], v[u]);
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const float* br = Bf + (int64_t)c[u] * ldb;
#pragma unroll
        for (int p = 0; p < NPASS; ++p) {
          if (fok[p]) ld_vec<VW>(br + p * L * VW, b[u][p]);
        }
      }
#pragma unroll
      for (int u = 0; u < U; ++u)
#pragma unroll
        for (int p = 0; p < NPASS; ++p)
          if (fok[p]) {
#pragma unroll
            for (int e = 0; e < VW; ++e) acc[p][e] = fmaf(v[u], b[u][p][e], acc[p][e]);
          }
    }
    for (; j < cnt; ++j) {
      int c; float v;
      bcast_entry<L>(cv, vv, g, j, c, v);
      const float* br = Bf + (int64_t)c * ldb;
      float b[NPASS][VW];
#pragma unroll
      for (int p = 0; p < NPASS; ++p)
        if (fok[p]) ld_vec<VW>(br + p * L * VW, b[p]);
#pragma unroll
      for (int p = 0; p < NPASS; ++p)
        if (fok[p]) {
#pragma unroll
          for (int e = 0; e < VW; ++e) acc[p][e] = fmaf(v, b[p][e], acc[p][e]);
        }
    }
  }

  if (row < n) {
#pragma unroll
    for (int p = 0; p < NPASS; ++p)
      if (fok[p]) st_vec<VW>(cr + p * L * VW, acc[p]);
  }
}

// ---------------------------------------------------------------------------------------
// Fast path: plan-driven, staged SpMM.
//
// A plan (dgnn_plan, include/dgnn_spmm.h) cuts the column space into `nstage` blocks. A
// *segment* is the run of one row's nonzeros inside one column block (contiguous, because CSR
// columns ascend). Stages are separate launches on one stream, each accumulating into C, so an
// output element still sees its nonzeros in ascending column order across stages. Inside a
// stage the segments are listed longest first (LPT), and a segment longer than the plan's
// piece length is cut into pieces whose sums go to `partial` and are added to C in piece order
// by merge_kernel right after the stage.
//
// One group of L lanes takes one segment and one 4*L-float feature chunk; B rows are read as
// float4 (global_load_dwordx4: B must have ldb % 4 == 0 and a 16-byte-aligned base, padding
// columns zero or anything finite -- they are never stored), C as VWC-float vectors.
//
// XCD-aware order: workgroups are dealt round-robin over the 8 XCDs (blockIdx % 8 labels the
// XCD group; each XCD has a private 4 MiB L2). The work index is laid out chunk-major and XCD
// group x takes the x-th contiguous eighth of it, so at any moment the workgroups resident on
// one XCD gather from one 256-byte feature chunk of one column block of B: the L2 working set
// is (columns per block) x 256 B instead of whole B rows.
// ---------------------------------------------------------------------------------------
typedef float f32x2 __attribute__((ext_vector_type(2)));
constexpr int SEG_MAX = 256;   // LDS strip of a 16-lane group, in entries (32-lane groups: 512, 8-lane groups: 128)

// C-side access helpers: a lane owns 4 consecutive floats of a row of C and touches them as VWC-float vectors
// while they lie inside [0, F); the vector that straddles F (a width that is no multiple of VWC, in a C with
// a wider leading dimension) goes element by element: nothing at or past column F is read or written.
template <int VWC>
__device__ __forceinline__ void c_load(const float* p, int fbase, int F, f32x2& a01, f32x2& a23) {
  if (VWC == 4 && fbase + 4 <= F) {
    const float4 t = *reinterpret_cast<const float4*>(p); a01 = {t.x, t.y}; a23 = {t.z, t.w};
    return;
  }
  if (VWC >= 2 && fbase + 2 <= F) { const float2 t = *reinterpret_cast<const float2*>(p); a01 = {t.x, t.y}; }
  else { if (fbase < F) a01.x = p[0]; if (fbase + 1 < F) a01.y = p[1]; }
  if (VWC >= 2 && fbase + 4 <= F) { const float2 t = *reinterpret_cast<const float2*>(p + 2); a23 = {t.x, t.y}; }
  else { if (fbase + 2 < F) a23.x = p[2]; if (fbase + 3 < F) a23.y = p[3]; }
}

template <int VWC>
__device__ __forceinline__ void c_store(float* p, int fbase, int F, f32x2 a01, f32x2 a23) {
  if (VWC == 4 && fbase + 4 <= F) {
    *reinterpret_cast<float4*>(p) = make_float4(a01.x, a01.y, a23.x, a23.y);
    return;
  }
  if (VWC >= 2 && fbase + 2 <= F) *reinterpret_cast<float2*>(p) = make_float2(a01.x, a01.y);
  else { if (fbase < F) p[0] = a01.x; if (fbase + 1 < F) p[1] = a01.y; }
  if (VWC >= 2 && fbase + 4 <= F) *reinterpret_cast<float2*>(p + 2) = make_float2(a23.x, a23.y);
  else { if (fbase + 2 < F) p[2] = a23.x; if (fbase + 3 < F) p[3] = a23.y; }
}

// One gather round: R entries of the group's LDS strip starting at j -> R rows of B in flight per lane,
// then R fused multiply-adds per accumulator, in entry order.
template <int R, bool WIDE>
__device__ __forceinline__ void gather_round(const uint2* ent, int j, const char* Bbytes, uint32_t ldb_bytes,
                                             uint32_t loff, f32x2& a01, f32x2& a23) {
  uint2 e[R];
  if constexpr (R == 1) {
    e[0] = ent[j];
  } else {
#pragma unroll
    for (int h = 0; h < R / 2; ++h) {                        // two entries per ds_read_b128
      const uint4 t = *reinterpret_cast<const uint4*>(&ent[j + 2 * h]);
      e[2 * h] = make_uint2(t.x, t.y);
      e[2 * h + 1] = make_uint2(t.z, t.w);
    }
  }
  float4 bb[R];
#pragma unroll
  for (int u = 0; u < R; ++u) {
    if constexpr (WIDE) bb[u] = *reinterpret_cast<const float4*>(Bbytes + ((int64_t)e[u].x * ldb_bytes + loff));
    else bb[u] = *reinterpret_cast<const float4*>(Bbytes + (e[u].x + loff));
  }
#pragma unroll
  for (int u = 0; u < R; ++u) {
    const float v = __uint_as_float(e[u].y);
    const f32x2 vv = {v, v};
    a01 = __builtin_elementwise_fma(vv, (f32x2){bb[u].x, bb[u].y}, a01);
    a23 = __builtin_elementwise_fma(vv, (f32x2){bb[u].z, bb[u].w}, a23);
  }
}

// Instruction budget of the inner loop. With the gathers L2-resident the kernel is bound by the texture-
// address unit (PMC: TA 86-88 % busy, ~21 TA cycles per wave-level global_load_dwordx4 against a floor of
// 16), so everything else has to stay out of its way: a strip of (col, val) entries is turned ONCE into
// (32-bit byte offset of the B row, alpha*val) pairs and parked in a group-private LDS strip; every lane
// of the group then reads two entries per ds_read_b128 at a group-uniform address (LDS broadcast), adds
// its own lane offset, and issues global_load_dwordx4 with the scalar base of B (saddr form). Per entry
// and lane: 1/2 LDS read + 1 add + 1 load + 2 packed FMAs.
template <int L, int U, int VWC, bool WIDE>
__global__ __launch_bounds__(256) void spmm_stage_kernel(
    const int4* __restrict__ seg, int32_t nseg, uint32_t nsegblocks,
    const int32_t* __restrict__ col, const float* __restrict__ val,
    const float* __restrict__ B, int64_t ldb, float* __restrict__ C, int64_t ldc,
    float* __restrict__ P, int64_t ldp, int32_t F, float alpha, uint32_t k, int fresh) {
  constexpr int G = 64 / L;
  constexpr int SEGM = SEG_MAX * L / 16;        // strip length: 16 entries per lane of the group
  static_assert(U % 2 == 0, "entries are read two per ds_read_b128");
  __shared__ uint2 ent_all[4][G][SEGM];         // (row offset or column, alpha*val) of the group's run
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int g = lane / L;
  const int li = lane % L;
  uint2* const ent = ent_all[wave][g];

  const uint32_t nb = gridDim.x, b = blockIdx.x;
  const uint32_t q = nb >> 3, r = nb & 7, xcd = b & 7;
  // k == 0 (experiment, variant 404-407): plain chunk-major order -- every XCD works on the same feature chunk at
  // the same time (the Infinity Cache then sees one chunk of B instead of all of them)
  const uint32_t w = k == 0 ? b : (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (b >> 3);
  const uint32_t chunk = w / nsegblocks;
  // Inside a chunk the segment blocks are listed longest first, and an XCD's share of the work index is
  // a contiguous range that may cover a fraction of a chunk -- which must not be "the long ones".
  // A chunk is therefore dealt into k classes (class j = blocks j, j + k, j + 2k, ...: same mix of long
  // and short runs in each; k = 8 / gcd(chunks, 8) makes every share a union of whole classes), and
  // the classes an XCD owns in one chunk are walked together, longest first.
  const uint32_t p = w - chunk * nsegblocks;
  uint32_t sb = p;
  if (k > 1) {
    const uint32_t qc = nsegblocks / k;                       // blocks per class (nsegblocks % k == 0)
    const uint32_t share = nb >> 3;                           // nb % 8 == 0 here
    const uint32_t v0 = xcd * share, c0 = chunk * nsegblocks;
    const uint32_t lo = (v0 > c0 ? v0 - c0 : 0u);
    const uint32_t hi = min(v0 + share - c0, nsegblocks);
    const uint32_t j0 = lo / qc, ncl = hi / qc - j0;
    const uint32_t t = p - lo;
    sb = (t / ncl) * k + j0 + t % ncl;
  }
  const int64_t sidx = ((int64_t)sb * 4 + wave) * G + g;
  const int fbase = (chunk * L + li) * 4;
  int4 s = make_int4(-1, 0, 0, -1);            // {row, k0, k1, out}: out < 0 -> accumulate into C[row]
  if (sidx < nseg) s = seg[sidx];
  const int kb = s.y, ke = s.z;
  // row -1: an unused slot of a list sized by an upper bound; an empty run (the first stage lists every row,
  // dgnn_plan.covers_rows) matters only to a product that overwrites C: it stores the row's zeros
  const bool live = s.x >= 0 && (ke > kb || fresh);
  const bool direct = s.w < 0;
  // lanes past the padded width (last chunk of a width that is no multiple of the chunk) gather nothing
  const bool gathers = fbase < (int)ldb;
  const uint32_t loff = (uint32_t)fbase * 4u;
  const uint32_t ldb_bytes = (uint32_t)ldb * 4u;
  const char* const Bbytes = reinterpret_cast<const char*>(B);

  float* const dst = direct ? (C + (int64_t)s.x * ldc + fbase) : (P + (int64_t)s.w * ldp + fbase);
  f32x2 a01 = {0.f, 0.f}, a23 = {0.f, 0.f};
  // pieces start from 0, direct runs from C -- unless C has no value yet (`fresh`: the first stage of a product
  // that overwrites C), in which case C is not read at all; an empty run (k0 == k1) then stores the row's zeros
  if (live && direct && !fresh) c_load<VWC>(dst, fbase, F, a01, a23);

  // A run is walked strip by strip (SEGM entries). Per strip --
  // Phase 1: the strip's (col, val) stream -- cold data, read once, HBM/Infinity-Cache latency -- is
  // fetched up front, all loads in flight together, and parked in LDS. Vector loads return in order, so
  // a cold load issued between gather rounds would stall every younger (L2-hit) gather behind it; this
  // way a strip pays that latency once. The strip is private to the group (one wave, in-order LDS), so
  // no s_barrier separates the phases or consecutive strips -- only wave-scope fences for the compiler.
#pragma unroll 1
  for (int sb0 = kb; sb0 < ke; sb0 += SEGM) {
    const int se = min(sb0 + SEGM, ke);
    const int len = se - sb0;
    if (len <= L) {                             // short run (sparse graphs): one entry per lane is all there is
      const int kk = min(sb0 + li, se - 1);
      const uint32_t c1 = (uint32_t)col[kk];
      const float v1 = val[kk];
      ent[li] = make_uint2(WIDE ? c1 : c1 * ldb_bytes, __float_as_uint(alpha * v1));
    } else
#pragma unroll 1
    for (int t0 = 0; t0 * L < len; t0 += 4) {   // 64 entries (8 loads per lane) in flight per trip
      uint32_t cx[4], vy[4];
#pragma unroll
      for (int t = 0; t < 4; ++t) {             // clamped, branch-free: tail lanes re-read the last entry
        const int kk = min(sb0 + li + (t0 + t) * L, se - 1);
        cx[t] = (uint32_t)col[kk];
        vy[t] = __float_as_uint(val[kk]);
      }
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const uint32_t x = WIDE ? cx[t] : cx[t] * ldb_bytes;
        ent[(li + (t0 + t) * L) & (SEGM - 1)] = make_uint2(x, __float_as_uint(alpha * __uint_as_float(vy[t])));
      }
    }
    // The strip is handed from the lanes that wrote it to the lanes that read it inside one wave: LDS is in
    // order per wave, but the compiler has to be told that these stores and the loads below communicate.
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    // Phase 2: gather rounds, U rows of B in flight per lane. Lanes past the padded width (last chunk
    // of a width that is no multiple of the chunk) sit the rounds out: fewer lanes for the address unit.
    if (gathers) {
      int j = 0;
#pragma unroll 1
      for (; j + U <= len; j += U) gather_round<U, WIDE>(ent, j, Bbytes, ldb_bytes, loff, a01, a23);
      // the leftover (< U entries) goes in rounds of 4, 2, 1: at most three memory latencies instead of one
      // per entry -- on sparse graphs most runs ARE leftover
      if constexpr (U > 4) if (j + 4 <= len) { gather_round<4, WIDE>(ent, j, Bbytes, ldb_bytes, loff, a01, a23); j += 4; }
      if constexpr (U > 2) if (j + 2 <= len) { gather_round<2, WIDE>(ent, j, Bbytes, ldb_bytes, loff, a01, a23); j += 2; }
#pragma unroll 1
      for (; j < len; ++j) gather_round<1, WIDE>(ent, j, Bbytes, ldb_bytes, loff, a01, a23);   // one trip for U = 2, 4, 8
    }
    // ... and the next strip's stores must not pass this strip's loads
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
  }

  if (live) {
    if (!direct) {                              // partial rows are ours: ldp % 4 == 0, 16-B aligned
      if (fbase < ldp) *reinterpret_cast<float4*>(dst) = make_float4(a01.x, a01.y, a23.x, a23.y);
    } else {
      c_store<VWC>(dst, fbase, F, a01, a23);
    }
  }
}


// ---------------------------------------------------------------------------------------
// Hot-tile variant of the stage kernel: "feature tiles of H_j staged in LDS" (north_star).
//
// The stage kernel above is bound by the texture-address / L1 path: every nonzero pulls its slice of a B row
// through it at 64 B/clk/CU, L2 hit or not. The only faster source on a CU is the LDS (256 B/clk/CU for
// ds_read_b128). A graph with hub columns lets it carry a large share of the reads: the plan names, per
// column-block stage, the T most referenced columns (`hot_cols`) and lists every run with its hot entries first
// (column replaced by the slot 0..T-1 in the tile) and its other entries after them. One workgroup of 16 waves
// per CU keeps the [T x CH] slice of those B rows for its feature chunk in LDS (128 KiB) and walks segment blocks
// persistently (the same chunk-major, XCD-aware, class-balanced order as the stage kernel, strided over the
// resident workgroups); per run a lane group first takes the hot entries out of the tile (ds_read_b128, no
// address unit), then gathers the others as the stage kernel does. The tile is reloaded only when the
// workgroup moves on to the next feature chunk.
//
// Accumulation order of an output element inside a stage: its hot nonzeros in ascending column order, then
// the others in ascending column order (one chain, one lane, fused multiply-adds: deterministic; the same
// bits as the strict chain over the plan's own (permuted) entry order, which is what the tests check).
// ---------------------------------------------------------------------------------------
constexpr int HOT_WAVES = 16;                 // waves per workgroup (1024 threads, one workgroup per CU)
constexpr int HOT_STRIP = 64;                 // entries of a run parked per lane group and trip
constexpr int HOT_TILE_FLOATS = 32768;        // 128 KiB of B rows

template <int R>
__device__ __forceinline__ void hot_round(const uint2* ent, int j, const char* tile_lane, f32x2& a01, f32x2& a23) {
  uint2 e[R];
  if constexpr (R == 1) {
    e[0] = ent[j];
  } else {
#pragma unroll
    for (int h = 0; h < R / 2; ++h) {
      const uint4 t = *reinterpret_cast<const uint4*>(&ent[j + 2 * h]);
      e[2 * h] = make_uint2(t.x, t.y);
      e[2 * h + 1] = make_uint2(t.z, t.w);
    }
  }
  float4 bb[R];
#pragma unroll
  for (int u = 0; u < R; ++u) bb[u] = *reinterpret_cast<const float4*>(tile_lane + e[u].x);
#pragma unroll
  for (int u = 0; u < R; ++u) {
    const float v = __uint_as_float(e[u].y);
    const f32x2 vv = {v, v};
    a01 = __builtin_elementwise_fma(vv, (f32x2){bb[u].x, bb[u].y}, a01);
    a23 = __builtin_elementwise_fma(vv, (f32x2){bb[u].z, bb[u].w}, a23);
  }
}

template <int L, int U, int VWC, bool WIDE>
__global__ __launch_bounds__(HOT_WAVES * 64) void spmm_hot_kernel(
    const int4* __restrict__ seg, const int32_t* __restrict__ seg_hot, int32_t nseg, uint32_t nsegblocks, uint32_t nbv,
    const int32_t* __restrict__ col, const float* __restrict__ val, const int32_t* __restrict__ hot_cols,
    const float* __restrict__ B, int64_t ldb, float* __restrict__ C, int64_t ldc,
    float* __restrict__ P, int64_t ldp, int32_t F, float alpha, uint32_t k, int fresh, int dbg) {
  constexpr int G = 64 / L;
  constexpr int CH = 4 * L;                      // floats of a feature chunk
  constexpr int T = HOT_TILE_FLOATS / CH;        // hot rows: 512 (L = 16) or 256 (L = 32)
  constexpr int PER = HOT_STRIP / L;             // strip entries loaded per lane and trip
  static_assert(U % 2 == 0, "entries are read two per ds_read_b128");
  __shared__ float4 tile[T * L];
  __shared__ uint2 ent_all[HOT_WAVES][G][HOT_STRIP];
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int g = lane / L;
  const int li = lane % L;
  uint2* const ent = ent_all[wave][g];
  const char* const tile_lane = reinterpret_cast<const char*>(tile) + li * 16;
  const uint32_t ldb_bytes = (uint32_t)ldb * 4u;
  const char* const Bbytes = reinterpret_cast<const char*>(B);

  // virtual block -> (feature chunk, segment block): the stage kernel's order over a virtual grid of nbv blocks
  // (nbv % 8 == 0), of which this workgroup takes blockIdx.x, blockIdx.x + gridDim.x, ... (gridDim.x % 8 == 0, so
  // all of them carry this workgroup's XCD label and walk that XCD's contiguous share front to back)
  const uint32_t q = nbv >> 3, xcd = blockIdx.x & 7;
  int cur_chunk = -1;
#pragma unroll 1
  for (uint32_t vb = blockIdx.x; vb < nbv; vb += gridDim.x) {
    const uint32_t w = xcd * q + (vb >> 3);
    const uint32_t chunk = w / nsegblocks;
    const uint32_t p = w - chunk * nsegblocks;
    uint32_t sb = p;
    if (k > 1) {
      const uint32_t qc = nsegblocks / k;
      const uint32_t v0 = xcd * q, c0 = chunk * nsegblocks;
      const uint32_t lo = (v0 > c0 ? v0 - c0 : 0u);
      const uint32_t hi = min(v0 + q - c0, nsegblocks);
      const uint32_t j0 = lo / qc, ncl = hi / qc - j0;
      const uint32_t t = p - lo;
      sb = (t / ncl) * k + j0 + t % ncl;
    }
    const int fbase = (chunk * L + li) * 4;
    const bool gathers = fbase < (int)ldb;       // lanes past the padded width read and gather nothing
    const uint32_t loff = (uint32_t)fbase * 4u;
    if ((int)chunk != cur_chunk) {               // uniform over the workgroup: vb depends on blockIdx only
      __syncthreads();                           // nobody still reads the previous chunk's tile
#pragma unroll 1
      for (int r = wave * G + g; r < T; r += HOT_WAVES * G) {
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (gathers) v = *reinterpret_cast<const float4*>(Bbytes + ((int64_t)hot_cols[r] * ldb_bytes + loff));
        tile[r * L + li] = v;
      }
      __syncthreads();
      cur_chunk = (int)chunk;
    }
    const int64_t sidx = ((int64_t)sb * HOT_WAVES + wave) * G + g;
    int4 s = make_int4(-1, 0, 0, -1);
    int kh = 0;
    if (sidx < nseg) { s = seg[sidx]; kh = seg_hot[sidx]; }
    const int kb = s.y, ke = s.z;
    const bool live = s.x >= 0 && (ke > kb || fresh);
    const bool direct = s.w < 0;
    float* const dst = direct ? (C + (int64_t)s.x * ldc + fbase) : (P + (int64_t)s.w * ldp + fbase);
    f32x2 a01 = {0.f, 0.f}, a23 = {0.f, 0.f};
    if (live && direct && !fresh) c_load<VWC>(dst, fbase, F, a01, a23);

    // ---- hot entries [kb, kh): (slot, val) -> (byte offset of the tile row, alpha*val), read back from the tile
#pragma unroll 1
    for (int sb0 = kb; sb0 < ((dbg & 2) ? kb : kh); sb0 += HOT_STRIP) {
      const int se = min(sb0 + HOT_STRIP, kh);
      const int len = se - sb0;
      uint32_t cx[PER], vy[PER];
#pragma unroll
      for (int t = 0; t < PER; ++t) {
        const int kk = min(sb0 + li + t * L, se - 1);
        cx[t] = (uint32_t)col[kk];
        vy[t] = __float_as_uint(val[kk]);
      }
#pragma unroll
      for (int t = 0; t < PER; ++t)
        ent[li + t * L] = make_uint2(cx[t] * (uint32_t)(CH * 4), __float_as_uint(alpha * __uint_as_float(vy[t])));
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      if (gathers) {
        int j = 0;
#pragma unroll 1
        for (; j + 8 <= len; j += 8) hot_round<8>(ent, j, tile_lane, a01, a23);
        if (j + 4 <= len) { hot_round<4>(ent, j, tile_lane, a01, a23); j += 4; }
        if (j + 2 <= len) { hot_round<2>(ent, j, tile_lane, a01, a23); j += 2; }
        if (j < len) hot_round<1>(ent, j, tile_lane, a01, a23);
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
    }
    // ---- the other entries [kh, ke): gathered through the vector-memory path, as in the stage kernel
#pragma unroll 1
    for (int sb0 = max(kh, kb); sb0 < ((dbg & 1) ? kb : ke); sb0 += HOT_STRIP) {
      const int se = min(sb0 + HOT_STRIP, ke);
      const int len = se - sb0;
      uint32_t cx[PER], vy[PER];
#pragma unroll
      for (int t = 0; t < PER; ++t) {
        const int kk = min(sb0 + li + t * L, se - 1);
        cx[t] = (uint32_t)col[kk];
        vy[t] = __float_as_uint(val[kk]);
      }
#pragma unroll
      for (int t = 0; t < PER; ++t)
        ent[li + t * L] = make_uint2(WIDE ? cx[t] : cx[t] * ldb_bytes, __float_as_uint(alpha * __uint_as_float(vy[t])));
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      if (gathers) {
        int j = 0;
#pragma unroll 1
        for (; j + U <= len; j += U) gather_round<U, WIDE>(ent, j, Bbytes, ldb_bytes, loff, a01, a23);
        if constexpr (U > 4) if (j + 4 <= len) { gather_round<4, WIDE>(ent, j, Bbytes, ldb_bytes, loff, a01, a23); j += 4; }
        if constexpr (U > 2) if (j + 2 <= len) { gather_round<2, WIDE>(ent, j, Bbytes, ldb_bytes, loff, a01, a23); j += 2; }
#pragma unroll 1
        for (; j < len; ++j) gather_round<1, WIDE>(ent, j, Bbytes, ldb_bytes, loff, a01, a23);
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
    }

    if (live) {
      if (!direct) {
        if (fbase < ldp) *reinterpret_cast<float4*>(dst) = make_float4(a01.x, a01.y, a23.x, a23.y);
      } else {
        c_store<VWC>(dst, fbase, F, a01, a23);
      }
    }
  }
}


// ---------------------------------------------------------------------------------------
// Row bundles: register reuse across rows that share columns.
//
// The stage kernel pays one gather of a B-row slice per nonzero, and the unit that serves gathers is what bounds
// it. Rows of similar length share many columns on graphs with hub columns, so the plan may tie R rows (R = 4 or
// 8, consecutive in the longest-first order of a stage) into a bundle: the union of their columns is walked once,
// each gathered slice is multiplied into up to R accumulators. Per union entry the plan stores the column and R
// values (0 where the row has no such nonzero: fma(0, b, acc) leaves acc as it is for finite b, so every row still
// sees exactly its own nonzeros in ascending column order -- the chain, and the bits, of the stage kernel).
//   desc   int32[(R + 4) * ndesc]: {row_0 .. row_(R-1) (-1: none), k0, k1, out, 0} per bundle piece, longest first;
//          out < 0: the rows accumulate into C; else piece sums go to partial rows out + r (merged per row afterwards)
//   ucol   int32[nu] union columns, uval float[nu * R] values, entry k at uval[k * R + r]
// ---------------------------------------------------------------------------------------
template <int R, int L, int U, int VWC, bool WIDE>
__global__ __launch_bounds__(256) void spmm_bundle_kernel(
    const int32_t* __restrict__ desc, int32_t ndesc, uint32_t nsegblocks,
    const int32_t* __restrict__ ucol, const float* __restrict__ uval,
    const float* __restrict__ B, int64_t ldb, float* __restrict__ C, int64_t ldc,
    float* __restrict__ P, int64_t ldp, int32_t F, float alpha, uint32_t k, int fresh) {
  constexpr int G = 64 / L;
  constexpr int SEGM = (R <= 4 ? 128 : 64) * L / 16;         // strip entries per lane group
  constexpr int VE = (R % 4 == 0) ? 4 : 2;                   // floats per value vector (R = 6: three float2)
  constexpr int NV = R / VE;
  using VT = typename Vec<VE>::T;
  static_assert(R % 2 == 0, "values are stored as float2 / float4 vectors");
  constexpr int PER = SEGM / L;                              // entries a lane loads per strip
  constexpr int DW = R + 4;
  __shared__ uint32_t off_all[4][G][SEGM];
  __shared__ VT val_all[4][G][SEGM * NV];
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int g = lane / L;
  const int li = lane % L;
  uint32_t* const off = off_all[wave][g];
  VT* const vals = val_all[wave][g];

  const uint32_t nb = gridDim.x, b = blockIdx.x;
  const uint32_t q = nb >> 3, xcd = b & 7;                   // nb % 8 == 0 (launcher)
  const uint32_t w = xcd * q + (b >> 3);
  const uint32_t chunk = w / nsegblocks;
  const uint32_t p = w - chunk * nsegblocks;
  uint32_t sb = p;
  if (k > 1) {
    const uint32_t qc = nsegblocks / k;
    const uint32_t v0 = xcd * q, c0 = chunk * nsegblocks;
    const uint32_t lo = (v0 > c0 ? v0 - c0 : 0u);
    const uint32_t hi = min(v0 + q - c0, nsegblocks);
    const uint32_t j0 = lo / qc, ncl = hi / qc - j0;
    const uint32_t t = p - lo;
    sb = (t / ncl) * k + j0 + t % ncl;
  }
  const int64_t didx = ((int64_t)sb * 4 + wave) * G + g;
  const int fbase = (chunk * L + li) * 4;
  const bool gathers = fbase < (int)ldb;
  const uint32_t loff = (uint32_t)fbase * 4u;
  const uint32_t ldb_bytes = (uint32_t)ldb * 4u;
  const char* const Bbytes = reinterpret_cast<const char*>(B);

  int rows[R];
  int kb = 0, ke = 0, out = -1;
#pragma unroll
  for (int r = 0; r < R; ++r) rows[r] = -1;
  if (didx < ndesc) {
    const int32_t* d = desc + didx * DW;
#pragma unroll
    for (int r = 0; r < R; ++r) rows[r] = d[r];
    kb = d[R]; ke = d[R + 1]; out = d[R + 2];
  }
  const bool direct = out < 0;
  f32x2 a01[R], a23[R];
#pragma unroll
  for (int r = 0; r < R; ++r) {
    a01[r] = {0.f, 0.f}; a23[r] = {0.f, 0.f};
    if (rows[r] >= 0 && direct && !fresh) c_load<VWC>(C + (int64_t)rows[r] * ldc + fbase, fbase, F, a01[r], a23[r]);
  }

#pragma unroll 1
  for (int sb0 = kb; sb0 < ke; sb0 += SEGM) {
    const int se = min(sb0 + SEGM, ke);
    const int len = se - sb0;
#pragma unroll 1
    for (int t = 0; t < PER && t * L < len; ++t) {
      const int kk = min(sb0 + li + t * L, se - 1);
      const uint32_t c1 = (uint32_t)ucol[kk];
      float v[R];
#pragma unroll
      for (int h = 0; h < NV; ++h) ld_vec<VE>(uval + (int64_t)kk * R + VE * h, *reinterpret_cast<float(*)[VE]>(&v[VE * h]));
      off[li + t * L] = WIDE ? c1 : c1 * ldb_bytes;
#pragma unroll
      for (int h = 0; h < NV; ++h) {
        float sv[VE];
#pragma unroll
        for (int e = 0; e < VE; ++e) sv[e] = alpha * v[VE * h + e];
        st_vec<VE>(reinterpret_cast<float*>(&vals[(li + t * L) * NV + h]), sv);
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    if (gathers) {
      int j = 0;
#pragma unroll 1
      for (; j + U <= len; j += U) {
        uint32_t o[U];
        float4 bb[U];
#pragma unroll
        for (int u = 0; u < U; ++u) o[u] = off[j + u];
#pragma unroll
        for (int u = 0; u < U; ++u) {
          if constexpr (WIDE) bb[u] = *reinterpret_cast<const float4*>(Bbytes + ((int64_t)o[u] * ldb_bytes + loff));
          else bb[u] = *reinterpret_cast<const float4*>(Bbytes + (o[u] + loff));
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
#pragma unroll
          for (int h = 0; h < NV; ++h) {
            float vr[VE];
            ld_vec<VE>(reinterpret_cast<const float*>(&vals[(j + u) * NV + h]), vr);
#pragma unroll
            for (int e = 0; e < VE; ++e) {
              const f32x2 vv = {vr[e], vr[e]};
              a01[VE * h + e] = __builtin_elementwise_fma(vv, (f32x2){bb[u].x, bb[u].y}, a01[VE * h + e]);
              a23[VE * h + e] = __builtin_elementwise_fma(vv, (f32x2){bb[u].z, bb[u].w}, a23[VE * h + e]);
            }
          }
        }
      }
#pragma unroll 1
      for (; j < len; ++j) {
        const uint32_t o = off[j];
        float4 bb;
        if constexpr (WIDE) bb = *reinterpret_cast<const float4*>(Bbytes + ((int64_t)o * ldb_bytes + loff));
        else bb = *reinterpret_cast<const float4*>(Bbytes + (o + loff));
#pragma unroll
        for (int h = 0; h < NV; ++h) {
          float vr[VE];
          ld_vec<VE>(reinterpret_cast<const float*>(&vals[j * NV + h]), vr);
#pragma unroll
          for (int e = 0; e < VE; ++e) {
            const f32x2 vv = {vr[e], vr[e]};
            a01[VE * h + e] = __builtin_elementwise_fma(vv, (f32x2){bb.x, bb.y}, a01[VE * h + e]);
            a23[VE * h + e] = __builtin_elementwise_fma(vv, (f32x2){bb.z, bb.w}, a23[VE * h + e]);
          }
        }
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
  }

#pragma unroll
  for (int r = 0; r < R; ++r) {
    if (rows[r] < 0) continue;
    if (!direct) {
      if (fbase < ldp) *reinterpret_cast<float4*>(P + (int64_t)(out + r) * ldp + fbase) = make_float4(a01[r].x, a01[r].y, a23[r].x, a23[r].y);
    } else if (ke > kb || fresh) {
      c_store<VWC>(C + (int64_t)rows[r] * ldc + fbase, fbase, F, a01[r], a23[r]);
    }
  }
}

// ---------------------------------------------------------------------------------------
// Dense tiles: register reuse across 32 rows where the block is dense enough to pay for the zeros.
//
// Graphs with hub rows and hub columns hold a large share of their nonzeros in a part of the matrix that is far
// from sparse (Reddit shape: rows taken 32 at a time in order of length, the columns that at least 6 of a tile's
// 32 rows reference hold 47 % of the nonzeros at 39 % density). There a gather per nonzero -- or per union entry of
// a 4-row bundle -- is the wrong currency: the plan (dgnn_dense_plan) lists, per tile of 32 rows, those columns
// and a dense [columns x 32] value array (0 where a row lacks the column), and one wave walks a piece of that list
// for one 128-float feature chunk: per column ONE coalesced 512-byte slice of the B row (a float2 per lane) feeds
// 32 rows x 2 packed FMAs whose A operands are wave-uniform and arrive through the scalar data path
// (s_load_dwordx16 straight into the SGPR operand of v_pk_fma_f32): no LDS, no lane traffic, 32 VALU instructions
// per vector load -- bound by the fp32 FMA rate, not by the address unit. No MFMA.
//
// Every output element is still one chain of fused multiply-adds in the list's (ascending column) order, zero
// weights leaving it as it is; pieces of a long list go to partial rows and are merged in piece order.
// The kernel accumulates into C (which the sparse remainder of the block has written before).
// ---------------------------------------------------------------------------------------
constexpr int DENSE_TR = 32;                  // rows of a tile = accumulator pairs per lane

__device__ __forceinline__ f32x2 c2_load(const float* p, int fbase, int F, bool vw2) {
  f32x2 a = {0.f, 0.f};
  if (vw2 && fbase + 2 <= F) { const float2 t = *reinterpret_cast<const float2*>(p); a = {t.x, t.y}; }
  else { if (fbase < F) a.x = p[0]; if (fbase + 1 < F) a.y = p[1]; }
  return a;
}

__device__ __forceinline__ void c2_store(float* p, int fbase, int F, bool vw2, f32x2 a) {
  if (vw2 && fbase + 2 <= F) *reinterpret_cast<float2*>(p) = make_float2(a.x, a.y);
  else { if (fbase < F) p[0] = a.x; if (fbase + 1 < F) p[1] = a.y; }
}

// Rows are paired: the SGPR pair {a[2j], a[2j+1]} that a scalar load leaves 64-bit aligned is the first operand of
// one v_pk_fma_f32 as it stands, the lane's B value is the broadcast one (op_sel on a VGPR costs nothing), so
// acc[j][e] = (C[row 2j][fbase + e], C[row 2j + 1][fbase + e]) and no scalar move is needed per value (the CU's one
// scalar unit could not keep up with 32 of them per column).
// W = floats per lane: 2 (a 128-float chunk per wave) or 1 (64 floats: feature blocks at most 64 floats wide).
template <bool ALPHA1, int W>
__device__ __forceinline__ void dense_fma(f32x2 (&acc)[DENSE_TR / 2][W], const float* __restrict__ a, f32x2 b, float alpha) {
#pragma unroll
  for (int j = 0; j < DENSE_TR / 2; ++j) {
    f32x2 a2 = {a[2 * j], a[2 * j + 1]};
    if (!ALPHA1) a2 = a2 * alpha;
    acc[j][0] = __builtin_elementwise_fma(a2, (f32x2){b.x, b.x}, acc[j][0]);
    if constexpr (W == 2) acc[j][1] = __builtin_elementwise_fma(a2, (f32x2){b.y, b.y}, acc[j][1]);
  }
}

template <bool ALPHA1, int DBG, int W>
__global__ __launch_bounds__(256) void spmm_dense_kernel(
    const int4* __restrict__ desc, int32_t nitems, int32_t nchunk,
    const int32_t* __restrict__ rows, const int32_t* __restrict__ kcol, const float* __restrict__ kval,
    const float* __restrict__ B, int64_t ldb, float* __restrict__ C, int64_t ldc,
    float* __restrict__ P, int64_t ldp, int32_t F, float alpha, int vw2) {
  constexpr int NP = DENSE_TR / 2;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int item = blockIdx.x * 4 + wave;                    // wave-uniform: everything below that does not depend
  if (item >= nitems) return;                                // on `lane` lives in SGPRs
  const int piece = item / nchunk, chunk = item - piece * nchunk;
  const int4 d = desc[piece];                                // {k0, k1, tile, out}; k1 - k0 is a multiple of 8 (lists are padded)
  const int fbase = (chunk * 64 + lane) * W;
  const bool active = fbase < (int)ldb;                      // lanes past the padded width compute on column 0, store nothing
  const uint64_t ldb_bytes = (uint64_t)ldb * 4u;
  const uint32_t loff = active ? (uint32_t)fbase * 4u : 0u;
  const char* const Bb = reinterpret_cast<const char*>(B);
  const int32_t* const rw = rows + (int64_t)d.z * DENSE_TR;
  const bool direct = d.w < 0;
  const bool v2 = W == 2 && vw2 != 0;

  f32x2 acc[NP][W];                                          // acc[j][e] = (C[row 2j][fbase + e], C[row 2j + 1][fbase + e])
#pragma unroll
  for (int j = 0; j < NP; ++j) {
    f32x2 c0 = {0.f, 0.f}, c1 = {0.f, 0.f};
    if (direct && active) {
      const int r0 = rw[2 * j], r1 = rw[2 * j + 1];
      if (r0 >= 0) c0 = c2_load(C + (int64_t)r0 * ldc + fbase, fbase, W == 2 ? F : min(F, fbase + 1), v2);
      if (r1 >= 0) c1 = c2_load(C + (int64_t)r1 * ldc + fbase, fbase, W == 2 ? F : min(F, fbase + 1), v2);
    }
    acc[j][0] = (f32x2){c0.x, c1.x};
    if constexpr (W == 2) acc[j][1] = (f32x2){c0.y, c1.y};
  }

  const int k0 = d.x, k1 = d.y;
  if (k0 < k1) {
    const uint32_t ldb_lo = (uint32_t)ldb_bytes;             // ldb * 4 < 2^32 (checked by the launcher)
    auto slice = [&](int k) -> f32x2 {                         // scalar row base (SALU) + 32-bit lane offset
      const uint32_t c = (DBG & 2) ? (uint32_t)(k & 63) : (uint32_t)kcol[k];
      const uint64_t off = ((uint64_t)__umulhi(c, ldb_lo) << 32) | (uint64_t)(c * ldb_lo);
      if constexpr (W == 2) return *reinterpret_cast<const f32x2*>(Bb + off + loff);
      else return (f32x2){*reinterpret_cast<const float*>(Bb + off + loff), 0.f};
    };
    // The value rows are a cold stream (HBM) and a scalar load cannot be issued far ahead (there are SGPRs for two
    // columns' 64 values only), so every trip would wait out a memory latency with nothing but the other waves' FMAs
    // to cover it (measured: 57 % of the FMA rate at 4 waves per SIMD). The wave therefore touches the 512 bytes of
    // the trip DENSE_PF columns ahead with one coalesced vector load that nobody waits for until a trip later: when the
    // scalar loads come, the lines are in L2.
    // Vector loads return in order, so a wait for a B slice also waits for every older load: the slices are fetched a
    // whole trip (4 columns) ahead and the touch goes out in the middle of a trip, which gives it a trip's time to land.
    constexpr int DENSE_PF = 16;                               // the touch runs this many columns ahead
    auto touch = [&](int k) -> f32x2 {                         // 512 B = the value rows of 4 columns, 8 B per lane
      return *reinterpret_cast<const f32x2*>(kval + (int64_t)min(k, k1 - 4) * DENSE_TR + lane * 2);
    };
    f32x2 pf = touch(k0 + DENSE_PF - 4);
    f32x2 b0 = slice(k0), b1 = slice(k0 + 1), b2 = slice(k0 + 2), b3 = slice(k0 + 3);
    f32x2 n0 = b0, n1 = b1, n2 = b2, n3 = b3;
#define DENSE_TRIP(K, B0, B1, B2, B3, N0, N1, N2, N3)                                         \
    {                                                                                         \
      const float* const a0 = kval + (int64_t)((DBG & 1) ? ((K) & 63) : (K)) * DENSE_TR;      \
      const int kn = min((K) + 4, k1 - 4);                                                    \
      dense_fma<ALPHA1, W>(acc, a0, B0, alpha);                                               \
      if (!(DBG & 8)) N0 = slice(kn);                                                         \
      dense_fma<ALPHA1, W>(acc, a0 + DENSE_TR, B1, alpha);                                    \
      if (!(DBG & 8)) N1 = slice(kn + 1);                                                     \
      asm volatile("" ::"v"(pf));          /* last trip's touch has landed (no instruction) */ \
      if (!(DBG & 4)) pf = touch((K) + DENSE_PF);                                             \
      dense_fma<ALPHA1, W>(acc, a0 + 2 * DENSE_TR, B2, alpha);                                \
      if (!(DBG & 8)) N2 = slice(kn + 2);                                                     \
      dense_fma<ALPHA1, W>(acc, a0 + 3 * DENSE_TR, B3, alpha);                                \
      if (!(DBG & 8)) N3 = slice(kn + 3);                                                     \
    }
#pragma unroll 1
    for (int k = k0; k < k1; k += 8) {                         // k1 - k0 is a multiple of 8 (lists are padded)
      DENSE_TRIP(k, b0, b1, b2, b3, n0, n1, n2, n3)
      DENSE_TRIP(k + 4, n0, n1, n2, n3, b0, b1, b2, b3)
    }
#undef DENSE_TRIP
    asm volatile("" ::"v"(pf));
  }

  if (!active) return;
#pragma unroll
  for (int j = 0; j < NP; ++j) {
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int row = rw[2 * j + h];
      if (row < 0) continue;
      f32x2 v = {h == 0 ? acc[j][0].x : acc[j][0].y, 0.f};
      if constexpr (W == 2) v.y = h == 0 ? acc[j][1].x : acc[j][1].y;
      if (direct) c2_store(C + (int64_t)row * ldc + fbase, fbase, W == 2 ? F : min(F, fbase + 1), v2, v);
      else if (fbase < ldp) {
        float* const pp = P + (int64_t)(d.w + 2 * j + h) * ldp + fbase;
        if constexpr (W == 2) *reinterpret_cast<float2*>(pp) = make_float2(v.x, v.y);
        else *pp = v.x;
      }
    }
  }
}

// ---------------------------------------------------------------------------------------
// Row strips: "feature tiles of H_j staged in LDS" (north_star), in the form that pays on graphs with hub columns.
//
// Every kernel above pulls one slice of a B row per nonzero (or per union entry of a few rows) through the texture-address
// unit, which is what bounds them (64 B/clk/CU, L2 hit or not). Here a workgroup owns a ROW TILE of 256 rows (rows ranked
// by length, 256 consecutive ranks: rows of similar length share the hub columns -- Reddit shape: the union of a tile's
// columns is 0.15 of its nonzeros) and one 256-float feature chunk. The tile's union column list is walked in ascending
// STRIPS of 64 columns; a strip's 64 slices of B (1 KiB each) go global -> LDS once, by LDS-DMA (global_load_lds_dwordx4:
// no register staging), double-buffered: strip s + 1 travels while strip s is multiplied. Each of the 16 waves owns 16 rows
// of the tile with all their accumulators in registers (a lane owns floats 2l, 2l+1 and 128+2l, 128+2l+1 of the chunk =
// 4 registers per row, 64 in all) and walks ITS rows' entries of the strip: one flat list of (slot, row, value) per
// (wave, strip), no zero padding of the matrix. Per entry and wave: two ds_read_b64 at slot * 2 KiB + lane * 8 (conflict
// free: the LDS image of a slice is the B row in its own order) and two v_pk_fma_f32. What makes a flat list possible is
// the accumulator ROW being data: the FMAs run in VGPR index mode (s_set_gpr_idx_on: M0[7:0] is added to the register
// number of dst and src2), so acc[row] += val * b is ONE instruction for a wave-uniform dynamic row.
//   Entries reach the wave through ONE coalesced vector load per strip (lane k holds entry k of the wave's block, fetched a
// whole strip ahead; vector memory returns in order, so the wait is counted -- scalar loads share a counter with the LDS and
// return out of order, which forbids running ahead of them) and are broadcast with v_readlane_b32 into an SGPR pair
// {slot << 11 | row * 4, value}: the pair is the first operand of the packed FMA as it stands (op_sel picks the value for
// both halves), its low byte is the register index, (pair.x & ~0x1ff) | lane base is the LDS address.
//   The inner loop is inline assembly on pinned registers (accumulators v[64:127], scratch v[24:47], s[36:43]): the compiler
// cannot express index mode, and a register tuple operand cannot be split into the pairs the instructions need.
//   Per row the entries come in ascending column order (strips ascend, a block is sorted by row then column) and the chain
// starts from what C holds: rows of uncut tiles carry the bits of the strict chain. A tile whose union list is longer than
// the plan's piece length is cut at strip boundaries into pieces (partial rows, merged per row in piece order).
// Pad entries (blocks are padded to whole groups of 4) carry the value -0.0 and the slot of an all-zero slice:
// fma(-0.0, +0.0, acc) == acc for every acc, signed zeros and NaN included -- nothing is changed by them.
// No MFMA, no atomics.
// ---------------------------------------------------------------------------------------
constexpr int STRIP_W = 64;                    // union columns per strip, at most (a piece may use 8, 16 or 32)
constexpr int STRIP_NW = 16;                   // waves per workgroup
constexpr int STRIP_RW = 16;                   // rows per wave
constexpr int STRIP_R = STRIP_NW * STRIP_RW;   // rows per tile
constexpr int STRIP_CH = 256;                  // floats per feature chunk
constexpr int STRIP_LDS_B = (STRIP_W + 1) * 2048;   // slot s of buffer q at (2 s + q) KiB; slot STRIP_W holds zeros
constexpr int STRIP_LINE = 32;                 // entries of a line of a wave's entry stream (256 B = one LDS-DMA instruction)
constexpr int STRIP_RING = 3;                  // lines a wave holds in LDS: in use, next, the one after
constexpr int STRIP_LDS = STRIP_LDS_B + STRIP_NW * STRIP_RING * STRIP_LINE * 8;

typedef float f32x32 __attribute__((ext_vector_type(32)));
typedef __attribute__((address_space(3))) void* lds_void_ptr;

// Multiply the line at LDS address `rb` (`ra` = rb + 4 * lane: lane l reads dword l) into the accumulators. The instruction
// stream comes from tools/gen_strip_asm.py (groups of 4 entries, software-pipelined across two register sets). A line is 64
// dwords: dword k = meta word of entry k (slot << 11 | row * 4 [| header]), dword 32 + k = its value. v_readlane_b32
// broadcasts the meta word into an SGPR whose low byte is the accumulator index and whose slot field, (meta & mask) | lane
// base, is the LDS address of the column's slice; the four values of a group reach every lane through one ds_read_b128 at a
// wave-uniform address and are the first operand of the packed FMAs from there (op_sel picks the half). The line's first
// entry carries the header: bits 20-23 the groups of 4 entries to take (1..8), bit 24 "last line of this strip". Returns bit
// 24; k0..k3 = dwords 28..31 of the line (the first line of a block keeps column ids there, and at most 7 groups of entries).
__device__ __forceinline__ uint32_t strip_line(f32x32& accA, f32x32& accB, uint32_t ra, uint32_t rb, uint32_t lb,
                                               int& k0, int& k1, int& k2, int& k3) {
  const uint32_t mask = 0x0003fe00u;
  uint32_t last;
  asm volatile(
#include "strip_line_asm.inc"
      : "+{v[64:95]}"(accA), "+{v[96:127]}"(accB), [last] "=&s"(last), [k0] "=&s"(k0), [k1] "=&s"(k1), [k2] "=&s"(k2), [k3] "=&s"(k3)
      : [ra] "v"(ra), [rb] "v"(rb), [mask] "v"(mask), [lb] "v"(lb)
      : "v14", "v15", "v16", "v17", "v18", "v19", "v20", "v21", "v24", "v25", "v26", "v27", "v28", "v32", "v33", "v34", "v35", "v36", "v37", "v38", "v39", "v40", "v41", "v42", "v43",
        "v44", "v45", "v46", "v47", "v48", "v49", "v50", "v51", "v52", "v53", "v54", "v55", "v56", "v57", "v58", "v59", "v60", "v61", "v62", "v63",
        "s36", "s37", "s38", "s39", "s40", "s41", "s42", "s43", "s44", "s46", "s47", "s48", "s49", "s50", "s51", "s52", "s53", "scc", "memory");
  return last;
}

#ifdef DGNN_EXPERIMENTS
// Timing experiments only (never in the product library): the same line walk with the slices' LDS reads or the packed FMAs
// left out (tools/gen_strip_asm.py noread | nofma). Same addresses, same waits on everything that carries an address.
__device__ __forceinline__ uint32_t strip_line_noread(f32x32& accA, f32x32& accB, uint32_t ra, uint32_t rb, uint32_t lb,
                                                      int& k0, int& k1, int& k2, int& k3) {
  const uint32_t mask = 0x0003fe00u;
  uint32_t last;
  asm volatile(
#include "strip_line_asm_noread.inc"
      : "+{v[64:95]}"(accA), "+{v[96:127]}"(accB), [last] "=&s"(last), [k0] "=&s"(k0), [k1] "=&s"(k1), [k2] "=&s"(k2), [k3] "=&s"(k3)
      : [ra] "v"(ra), [rb] "v"(rb), [mask] "v"(mask), [lb] "v"(lb)
      : "v14", "v15", "v16", "v17", "v18", "v19", "v20", "v21", "v24", "v25", "v26", "v27", "v28", "v32", "v33", "v34", "v35", "v36", "v37", "v38", "v39", "v40", "v41", "v42", "v43",
        "v44", "v45", "v46", "v47", "v48", "v49", "v50", "v51", "v52", "v53", "v54", "v55", "v56", "v57", "v58", "v59", "v60", "v61", "v62", "v63",
        "s36", "s37", "s38", "s39", "s40", "s41", "s42", "s43", "s44", "s46", "s47", "s48", "s49", "s50", "s51", "s52", "s53", "scc", "memory");
  return last;
}
__device__ __forceinline__ uint32_t strip_line_nofma(f32x32& accA, f32x32& accB, uint32_t ra, uint32_t rb, uint32_t lb,
                                                     int& k0, int& k1, int& k2, int& k3) {
  const uint32_t mask = 0x0003fe00u;
  uint32_t last;
  asm volatile(
#include "strip_line_asm_nofma.inc"
      : "+{v[64:95]}"(accA), "+{v[96:127]}"(accB), [last] "=&s"(last), [k0] "=&s"(k0), [k1] "=&s"(k1), [k2] "=&s"(k2), [k3] "=&s"(k3)
      : [ra] "v"(ra), [rb] "v"(rb), [mask] "v"(mask), [lb] "v"(lb)
      : "v14", "v15", "v16", "v17", "v18", "v19", "v20", "v21", "v24", "v25", "v26", "v27", "v28", "v32", "v33", "v34", "v35", "v36", "v37", "v38", "v39", "v40", "v41", "v42", "v43",
        "v44", "v45", "v46", "v47", "v48", "v49", "v50", "v51", "v52", "v53", "v54", "v55", "v56", "v57", "v58", "v59", "v60", "v61", "v62", "v63",
        "s36", "s37", "s38", "s39", "s40", "s41", "s42", "s43", "s44", "s46", "s47", "s48", "s49", "s50", "s51", "s52", "s53", "scc", "memory");
  return last;
}
#endif

template <int LV>                                 // 0: the product's line walk; 1 / 2: timing experiments (-DDGNN_EXPERIMENTS only)
__global__ __launch_bounds__(STRIP_NW * 64) void spmm_strip_kernel(
    const int32_t* __restrict__ desc, int32_t npiece, const int32_t* __restrict__ rows, const int32_t* __restrict__ ucol,
    const uint32_t* __restrict__ ent, const int64_t* __restrict__ ebase,
    const float* __restrict__ B, int64_t ldb, float* __restrict__ C, int64_t ldc,
    float* __restrict__ P, int64_t ldp, int32_t F, int vw2, int dbg, int fresh) {
  __shared__ __align__(16) unsigned char lds[STRIP_LDS];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  // chunk-major: the whole chip works on one 256-float chunk of B at a time (1 KiB x columns: what the Infinity Cache
  // and the L2s then hold), pieces longest first inside a chunk
  const int chunk = blockIdx.x / npiece, piece = blockIdx.x - chunk * npiece;
  const int32_t* const d = desc + (int64_t)piece * 8;          // {tile, k0, k1, out, strips, columns per strip, 0, 0}
  const int tile = d[0], k0 = d[1], k1 = d[2], out = d[3], nsp = d[4], Wt = d[5];
  const bool direct = out < 0;
  const int f0 = chunk * STRIP_CH + lane * 2, f1 = f0 + 128;   // the lane's two float pairs
  const bool act0 = f0 < (int)ldb, act1 = f1 < (int)ldb;       // ldb is a multiple of 32: a pair is inside the row or outside
  const bool v2 = vw2 != 0;
  const int32_t* const rw = rows + (int64_t)tile * STRIP_R;

  if (threadIdx.x < 128)                                       // the all-zero slice pad entries point at (both buffers)
    *reinterpret_cast<float4*>(lds + STRIP_W * 2048 + threadIdx.x * 16) = make_float4(0.f, 0.f, 0.f, 0.f);

  f32x32 accA, accB;                                           // row j of the wave: registers 4 j .. 4 j + 3 = {lo.x, lo.y, hi.x, hi.y}
#pragma unroll
  for (int j = 0; j < STRIP_RW; ++j) {
    f32x2 lo = {0.f, 0.f}, hi = {0.f, 0.f};
    if (direct && !fresh) {                                    // fresh: the tile's rows have no value yet -- C is written, not read
      const int row = rw[j * STRIP_NW + wave];
      if (row >= 0) {
        if (act0) lo = c2_load(C + (int64_t)row * ldc + f0, f0, F, v2);
        if (act1) hi = c2_load(C + (int64_t)row * ldc + f1, f1, F, v2);
      }
    }
    if (j < 8) { accA[4 * j] = lo.x; accA[4 * j + 1] = lo.y; accA[4 * j + 2] = hi.x; accA[4 * j + 3] = hi.y; }
    else { accB[4 * j - 32] = lo.x; accB[4 * j - 31] = lo.y; accB[4 * j - 30] = hi.x; accB[4 * j - 29] = hi.y; }
  }
  __syncthreads();                                             // the zero slice is in place; C has been read

  // From here on every vector-memory operation of the wave is an LDS-DMA (nothing lands in a register, so the compiler
  // inserts no waits of its own) and the waits are counted by hand: vector memory completes in order, a step ends by
  // requesting the entry line two steps ahead (one instruction), and the next step starts with vmcnt(1) -- everything
  // but that youngest request has landed: the slices of the strip about to be multiplied and the line about to be read.
  // The wave's entry stream is a sequence of 256-byte lines (32 entries; a block = the entries of one strip = whole lines),
  // so the requests simply walk it; what a line holds is in its header. Column ids come through the scalar cache
  // (wave-uniform, requested a strip ahead).
  // every lane moves 16 bytes of the slice; lanes past the padded width of the last chunk re-read the row's last 16 bytes
  // instead (in bounds; what lands in their part of the LDS image feeds accumulator lanes that are never stored), so the
  // requests need no per-lane predicate: the conditions below are wave-uniform and cost scalar branches only
  const int64_t lane_off = min((int64_t)chunk * (STRIP_CH * 4) + lane * 16, (int64_t)ldb * 4 - 16);
  const char* const Bb = reinterpret_cast<const char*>(B) + lane_off;
  const bool dma = !(dbg & 2);                                 // (dbg: timing experiments)
  const uint64_t ldb_bytes = (uint64_t)ldb * 4u;
  const int wslot = wave * 4;                                  // wave w moves slots 4 w .. 4 w + 3 of a strip
  const int kw = k0 + wslot;
  const bool wide = wslot + 3 < Wt;                            // this wave has four slots in every strip of the piece
  // (one 1-KiB instruction per slice: lane l copies floats 4l .. 4l+3, so the LDS image is the slice in its own order)
#define STRIP_DMA(C, OFF) __builtin_amdgcn_global_load_lds((const void*)(Bb + (uint64_t)(uint32_t)(C) * ldb_bytes), (lds_void_ptr)(dst + (OFF)), 16, 0, 0)
#define STRIP_FILL(S, C0, C1, C2, C3)                                                                                   \
  if (dma) {                                                                                                            \
    const int kk = kw + (S) * Wt;                                                                                       \
    unsigned char* const dst = lds + (2 * wslot + ((S) & 1)) * 1024;                                                    \
    if (wide && kk + 3 < k1) {                                 /* the common case: no further tests */                  \
      STRIP_DMA(C0, 0); STRIP_DMA(C1, 2048); STRIP_DMA(C2, 4096); STRIP_DMA(C3, 6144);                                  \
    } else {                                                                                                            \
      if (wslot + 0 < Wt && kk + 0 < k1) STRIP_DMA(C0, 0);                                                              \
      if (wslot + 1 < Wt && kk + 1 < k1) STRIP_DMA(C1, 2048);                                                           \
      if (wslot + 2 < Wt && kk + 2 < k1) STRIP_DMA(C2, 4096);                                                           \
      if (wslot + 3 < Wt && kk + 3 < k1) STRIP_DMA(C3, 6144);                                                           \
    }                                                                                                                   \
  }
#define STRIP_COLS(S, C0, C1, C2, C3)                                                                                   \
  {                                                                                                                     \
    const int kk = kw + (S) * Wt;                              /* clamped: slots nobody uses repeat a column */          \
    C0 = ucol[min(max(kk + 0, k0), k1 - 1)]; C1 = ucol[min(max(kk + 1, k0), k1 - 1)];                                   \
    C2 = ucol[min(max(kk + 2, k0), k1 - 1)]; C3 = ucol[min(max(kk + 3, k0), k1 - 1)];                                   \
  }
  const uint32_t* lp = ent + ebase[(int64_t)piece * STRIP_NW + wave] * (STRIP_LINE * 2) + lane;   // next line to request, this lane's dword
  unsigned char* const ring = lds + STRIP_LDS_B + wave * (STRIP_RING * STRIP_LINE * 8);
  const uint32_t ring_rb = (uint32_t)(STRIP_LDS_B + wave * (STRIP_RING * STRIP_LINE * 8));
  int c0, c1, c2, c3;                                          // columns of the wave's four slots of the next strip to fill
  STRIP_COLS(0, c0, c1, c2, c3)
  STRIP_FILL(0, c0, c1, c2, c3)
  STRIP_COLS(1, c0, c1, c2, c3)                                // (the only scalar loads: later ids ride in the entry lines)
  __builtin_amdgcn_global_load_lds((const void*)lp, (lds_void_ptr)(ring), 4, 0, 0);
  lp += STRIP_LINE * 2;
  __builtin_amdgcn_global_load_lds((const void*)lp, (lds_void_ptr)(ring + STRIP_LINE * 8), 4, 0, 0);
  lp += STRIP_LINE * 2;
  int rs = 0, ws = 2;                                          // ring slot to read / to request into
  int s = 0;
  uint32_t first = 1;
#pragma unroll 1
  while (s < nsp) {
    asm volatile("s_waitcnt vmcnt(1)" ::: "memory");            // all but the youngest line request have landed
    if (first) {
      if (!(dbg & 4)) __builtin_amdgcn_s_barrier();             // everybody's slices of s are in; everybody is done with s - 1
      if (s + 1 < nsp) STRIP_FILL(s + 1, c0, c1, c2, c3)
    }
    asm volatile("" ::: "memory");
    __builtin_amdgcn_global_load_lds((const void*)lp, (lds_void_ptr)(ring + ws * (STRIP_LINE * 8)), 4, 0, 0);
    lp += STRIP_LINE * 2;
    ws = ws == STRIP_RING - 1 ? 0 : ws + 1;
    const uint32_t lb = (uint32_t)lane * 8u + (uint32_t)(s & 1) * 1024u;
    int n0, n1, n2, n3;
    const uint32_t rb = ring_rb + rs * (STRIP_LINE * 8);
    uint32_t last;
#ifdef DGNN_EXPERIMENTS
    if constexpr (LV == 1) last = strip_line_noread(accA, accB, rb + (uint32_t)lane * 4u, rb, lb, n0, n1, n2, n3);
    else if constexpr (LV == 2) last = strip_line_nofma(accA, accB, rb + (uint32_t)lane * 4u, rb, lb, n0, n1, n2, n3);
    else
#endif
    last = strip_line(accA, accB, rb + (uint32_t)lane * 4u, rb, lb, n0, n1, n2, n3);
    if (first) { c0 = n0; c1 = n1; c2 = n2; c3 = n3; }          // a block's first line names the columns of strip s + 2
    rs = rs == STRIP_RING - 1 ? 0 : rs + 1;
    first = last;
    s += (int)last;
  }
#undef STRIP_FILL
#undef STRIP_DMA
#undef STRIP_COLS
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");              // the last requests land before the LDS goes back

#pragma unroll
  for (int j = 0; j < STRIP_RW; ++j) {
    const int i = j * STRIP_NW + wave;
    const int row = rw[i];
    if (row < 0) continue;
    f32x2 lo, hi;
    if (j < 8) { lo = (f32x2){accA[4 * j], accA[4 * j + 1]}; hi = (f32x2){accA[4 * j + 2], accA[4 * j + 3]}; }
    else { lo = (f32x2){accB[4 * j - 32], accB[4 * j - 31]}; hi = (f32x2){accB[4 * j - 30], accB[4 * j - 29]}; }
    if (direct) {
      if (act0) c2_store(C + (int64_t)row * ldc + f0, f0, F, v2, lo);
      if (act1) c2_store(C + (int64_t)row * ldc + f1, f1, F, v2, hi);
    } else {
      float* const pp = P + (int64_t)(out + i) * ldp;
      if (f0 < ldp) *reinterpret_cast<float2*>(pp + f0) = make_float2(lo.x, lo.y);
      if (f1 < ldp) *reinterpret_cast<float2*>(pp + f1) = make_float2(hi.x, hi.y);
    }
  }
}

// C[row] += partial[pbase] + partial[pbase+1] + ...  (fixed order). One wave per (cut run, 256-float chunk); a lane owns
// one float4 column of the (16-B aligned, padded) partial rows and keeps 4 piece loads in flight (16 measured slower:
// the merges are thousands of short runs, bound by their traffic, not by a chain of round trips).
template <int VWC>
__global__ __launch_bounds__(256) void merge_kernel(
    const int4* __restrict__ cut, int32_t ncut, const float* __restrict__ P, int64_t ldp,
    float* __restrict__ C, int64_t ldc, int32_t F, int fresh) {
  const int run = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (run >= ncut) return;
  const int4 c4 = cut[run];                    // {row, pbase, count, piece stride - 1}
  if (c4.z <= 0) return;                        // unused slot (list sized by an upper bound)
  const int lane = threadIdx.x & 63;
  const int64_t pst = (int64_t)(c4.w + 1) * ldp;   // partial rows of consecutive pieces are c4.w + 1 apart (row bundles: R)
  // blockIdx.y: the 256-float chunk of the row (one chunk per wave: the chunks of a run are independent chains of loads)
  for (int f = blockIdx.y * 256 + lane * 4; f < min(F, (int)(blockIdx.y + 1) * 256); f += 256) {
    float* c = C + (int64_t)c4.x * ldc + f;
    f32x2 a01 = {0.f, 0.f}, a23 = {0.f, 0.f};
    if (!fresh) c_load<VWC>(c, f, F, a01, a23);            // fresh: the row has no value yet (see the stage kernel)
    const float* p = P + (int64_t)c4.y * ldp + f;
    int s = 0;
    for (; s + 4 <= c4.z; s += 4) {
      float4 t[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) t[u] = *reinterpret_cast<const float4*>(p + (int64_t)(s + u) * pst);
#pragma unroll
      for (int u = 0; u < 4; ++u) { a01.x += t[u].x; a01.y += t[u].y; a23.x += t[u].z; a23.y += t[u].w; }
    }
    for (; s < c4.z; ++s) {
      const float4 t = *reinterpret_cast<const float4*>(p + (int64_t)s * pst);
      a01.x += t.x; a01.y += t.y; a23.x += t.z; a23.y += t.w;
    }
    c_store<VWC>(c, f, F, a01, a23);
  }
}

static void launch_merge(int vwc, const int4* cutp, int32_t nc, const float* partial, int64_t ldp, float* C, int64_t ldc,
                         int32_t F, int fresh, hipStream_t s) {
  const dim3 grid((unsigned)((nc + 3) / 4), (unsigned)((F > 0 ? F + 255 : 256) / 256));
  if (vwc == 4) hipLaunchKernelGGL(merge_kernel<4>, grid, dim3(256), 0, s, cutp, nc, partial, ldp, C, ldc, F, fresh);
  else if (vwc == 2) hipLaunchKernelGGL(merge_kernel<2>, grid, dim3(256), 0, s, cutp, nc, partial, ldp, C, ldc, F, fresh);
  else hipLaunchKernelGGL(merge_kernel<1>, grid, dim3(256), 0, s, cutp, nc, partial, ldp, C, ldc, F, fresh);
}

// ---------------------------------------------------------------------------------------
// Index / layout helpers.
// ---------------------------------------------------------------------------------------
// coo2csr: thread k closes every row boundary between row[k-1] and row[k].
__global__ __launch_bounds__(256) void coo2csr_kernel(const int32_t* __restrict__ row, int64_t nnz,
                                                      int32_t n, int32_t* __restrict__ rowptr) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; k <= nnz; k += stride) {
    const int32_t lo = (k == 0) ? -1 : row[k - 1];
    const int32_t hi = (k == nnz) ? n : row[k];
    for (int32_t r = lo + 1; r <= hi; ++r) rowptr[r] = (int32_t)k;
  }
}

__global__ __launch_bounds__(256) void idx_narrow_kernel(const int64_t* __restrict__ src, int64_t count,
                                                         int32_t* __restrict__ dst) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; k < count; k += stride)
    dst[k] = (int32_t)src[k];
}

// C = beta * C (beta == 0 writes zeros without reading).
__global__ __launch_bounds__(256) void scale_rows_kernel(float* __restrict__ C, int64_t ldc, int32_t n,
                                                         int32_t F, float beta) {
  const int64_t total = (int64_t)n * F;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += stride) {
    float* p = C + (t / F) * ldc + (t % F);
    *p = (beta == 0.f) ? 0.f : beta * *p;
  }
}

// dst[r, 0:F] = src[r, 0:F]; dst[r, F:ldd] = 0. One float4 of dst per thread (ldd % 4 == 0).
__global__ __launch_bounds__(256) void pad_rows_kernel(const float* __restrict__ src, int64_t lds,
                                                       float* __restrict__ dst, int64_t ldd,
                                                       int32_t n, int32_t F) {
  const int64_t q4 = ldd / 4;
  const int64_t total = (int64_t)n * q4;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += stride) {
    const int64_t r = t / q4;
    const int f = (int)(t - r * q4) * 4;
    const float* s = src + r * lds + f;
    float o[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) o[e] = (f + e < F) ? s[e] : 0.f;
    st_vec<4>(dst + r * ldd + f, o);
  }
}

// dst[r, 0:F] = src[idx[r], 0:F]; dst[r, F:ldd] = 0. One float4 of dst per thread; a row of dst is
// written by consecutive lanes (coalesced 16-B stores), the source row is read the same way.
__global__ __launch_bounds__(256) void gather_rows_kernel(const float* __restrict__ src, int64_t lds,
                                                          const int32_t* __restrict__ idx,
                                                          float* __restrict__ dst, int64_t ldd,
                                                          int32_t k, int32_t F, bool src_vec4) {
  const int64_t q4 = ldd / 4;
  const int64_t total = (int64_t)k * q4;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += stride) {
    const int64_t r = t / q4;
    const int f = (int)(t - r * q4) * 4;
    const float* s = src + (int64_t)idx[r] * lds + f;
    float o[4];
    if (src_vec4 && f + 4 <= F) {
      ld_vec<4>(s, o);
    } else {
#pragma unroll
      for (int e = 0; e < 4; ++e) o[e] = (f + e < F) ? s[e] : 0.f;
    }
    st_vec<4>(dst + r * ldd + f, o);
  }
}

// Work plan of the stage kernel for a block nobody planned, built on the device in three small passes over
// the rows -- the same plan shape a host-built dgnn_plan has: `ns` column-block stages; per stage one
// segment per non-empty (row, column block) run, runs longer than `piece` cut into pieces (-> partial rows
// -> merge); inside a stage everything grouped by length class (len_class below), longest class
// first, so long runs start early and the runs that share a wave have similar lengths. The lists are
// sized by upper bounds (the host never learns the counts): unused segment slots keep row = -1, unused
// cut slots count = 0. Header `hdr`: per stage PLAN_HDR_STAGE ints = [0, NCLS) class counts, [NCLS, 2 NCLS)
// write cursors, [2 NCLS] cut runs; then [PLAN_HDR_NPART] partial rows. Positions come from atomic counters, so the
// list order varies from run to run; the result does not (a run, or a piece, is one chain with one
// writer; pieces are merged in piece order).
constexpr int PLAN_MAX_STAGES = 16;
constexpr int PLAN_NCLS = 256;              // length classes: bit length and the three bits below the leading one
constexpr int PLAN_HDR_STAGE = 2 * PLAN_NCLS + 16;
constexpr int PLAN_HDR_NPART = PLAN_MAX_STAGES * PLAN_HDR_STAGE;
constexpr int PLAN_HDR_INTS = PLAN_HDR_NPART + 16;

// Runs of one class differ in length by at most 12.5 % (the groups of a wave finish together); classes are
// ordered by length.
__device__ __forceinline__ int len_class(int len) {
  if (len <= 0) return 0;
  const int bits = 32 - __clz(len);                                   // 1 .. 31
  const int frac = bits >= 4 ? (len >> (bits - 4)) & 7 : (len << (4 - bits)) & 7;
  return bits * 8 + frac;                                             // 8 .. 255
}

// first position in [lo, hi) whose column is >= bound (columns of a row ascend)
__device__ __forceinline__ int col_lower_bound(const int32_t* __restrict__ col, int lo, int hi, int bound) {
  while (lo < hi) {
    const int mid = (lo + hi) >> 1;
    if (col[mid] < bound) lo = mid + 1; else hi = mid;
  }
  return lo;
}

__global__ __launch_bounds__(256) void plan_hist_kernel(const int32_t* __restrict__ rowptr, const int32_t* __restrict__ col,
                                                        int32_t n, int ns, int cbw, int piece, int* __restrict__ hdr) {
  __shared__ int local[PLAN_MAX_STAGES][PLAN_NCLS];
  for (int t = threadIdx.x; t < ns * PLAN_NCLS; t += 256) local[t / PLAN_NCLS][t % PLAN_NCLS] = 0;
  __syncthreads();
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    int a = rowptr[i];
    const int k1 = rowptr[i + 1];
    // a row with nothing in the first column block is listed there as an empty run: the first stage then
    // stores every row (dgnn_plan.covers_rows)
    if (a >= k1 || (ns > 1 && col[a] >= cbw)) atomicAdd(&local[0][0], 1);
    for (int st = 0; st < ns && a < k1; ++st) {
      const int b = (st == ns - 1) ? k1 : col_lower_bound(col, a, k1, (st + 1) * cbw);
      const int len = b - a;
      if (len > 0 && len <= piece) {
        atomicAdd(&local[st][len_class(len)], 1);
      } else if (len > piece) {
        const int full = len / piece, rem = len - full * piece;
        atomicAdd(&local[st][len_class(piece)], full);
        if (rem) atomicAdd(&local[st][len_class(rem)], 1);
      }
      a = b;
    }
  }
  __syncthreads();
  for (int t = threadIdx.x; t < ns * PLAN_NCLS; t += 256) {
    const int v = local[t / PLAN_NCLS][t % PLAN_NCLS];
    if (v) atomicAdd(&hdr[(t / PLAN_NCLS) * PLAN_HDR_STAGE + t % PLAN_NCLS], v);
  }
}

__global__ void plan_cursor_kernel(int* __restrict__ hdr, int ns) {
  const int st = threadIdx.x;
  if (st < ns) {
    int* h = hdr + st * PLAN_HDR_STAGE;
    int at = 0;
    for (int c = PLAN_NCLS - 1; c >= 0; --c) { h[PLAN_NCLS + c] = at; at += h[c]; }
  }
}

__global__ __launch_bounds__(256) void plan_fill_kernel(const int32_t* __restrict__ rowptr, const int32_t* __restrict__ col,
                                                        int32_t n, int ns, int cbw, int piece, int* __restrict__ hdr,
                                                        int4* __restrict__ seg, int64_t seg_stride,
                                                        int4* __restrict__ cut, int64_t cut_stride) {
  // A tile of 256 rows ranks its uncut runs per (stage, class) in LDS and reserves their slots with one
  // global atomic per (stage, class): millions of rows hammering a few global counters one by one cost
  // more than the SpMM itself.
  __shared__ int lcnt[PLAN_MAX_STAGES][PLAN_NCLS], lbase[PLAN_MAX_STAGES][PLAN_NCLS];
  __shared__ int lend[PLAN_MAX_STAGES][256];                 // end of the run of (stage, thread)
  __shared__ unsigned short lrank[PLAN_MAX_STAGES][256];
  for (int64_t t0 = (int64_t)blockIdx.x * 256; t0 < n; t0 += (int64_t)gridDim.x * 256) {
    for (int t = threadIdx.x; t < ns * PLAN_NCLS; t += 256) lcnt[t / PLAN_NCLS][t % PLAN_NCLS] = 0;
    __syncthreads();
    const int64_t i = t0 + threadIdx.x;
    const bool valid = i < n;
    const int k0 = valid ? rowptr[i] : 0, k1 = valid ? rowptr[i + 1] : 0;
    int a = k0;
    for (int st = 0; st < ns; ++st) {
      const int b = (a >= k1) ? k1 : (st == ns - 1) ? k1 : col_lower_bound(col, a, k1, (st + 1) * cbw);
      lend[st][threadIdx.x] = b;
      const int len = b - a;
      if (valid && (len > 0 || st == 0) && len <= piece)
        lrank[st][threadIdx.x] = (unsigned short)atomicAdd(&lcnt[st][len_class(len)], 1);
      a = b;
    }
    __syncthreads();
    for (int t = threadIdx.x; t < ns * PLAN_NCLS; t += 256) {
      const int st = t / PLAN_NCLS, c = t % PLAN_NCLS;
      if (lcnt[st][c]) lbase[st][c] = atomicAdd(&hdr[st * PLAN_HDR_STAGE + PLAN_NCLS + c], lcnt[st][c]);
    }
    __syncthreads();
    a = k0;
    for (int st = 0; st < ns; ++st) {
      const int b = lend[st][threadIdx.x];
      const int len = b - a;
      int4* const sseg = seg + st * seg_stride;
      if (valid && (len > 0 || st == 0) && len <= piece) {      // stage 0 lists empty runs too
        sseg[lbase[st][len_class(len)] + lrank[st][threadIdx.x]] = make_int4((int)i, a, b, -1);
      } else if (len > piece) {
        int* const h = hdr + st * PLAN_HDR_STAGE;
        const int np = (len + piece - 1) / piece;
        const int pbase = atomicAdd(&hdr[PLAN_HDR_NPART], np);
        cut[st * cut_stride + atomicAdd(&h[2 * PLAN_NCLS], 1)] = make_int4((int)i, pbase, np, 0);
        for (int p = 0; p < np; ++p) {
          const int pa = a + p * piece, pb = min(pa + piece, b);
          sseg[atomicAdd(&h[PLAN_NCLS + len_class(pb - pa)], 1)] = make_int4((int)i, pa, pb, pbase + p);
        }
      }
      a = b;
    }
    __syncthreads();
  }
}

// The host-side rules of spmm_cpp (default_stage_count, default_piece_len), for plans built here.
int auto_stage_count(int64_t n, int64_t m, int64_t nnz, int64_t ldb) {
  if (n <= 0 || nnz <= 0) return 1;
  const int64_t budget = (ldb > 64) ? (16ll << 20) : (32ll << 20);
  const int64_t by_cache = (m * 256 + budget - 1) / budget;
  const int64_t by_density = nnz / n / 24 > 1 ? nnz / n / 24 : 1;
  int64_t ns = by_cache < by_density ? by_cache : by_density;
  if (ns > PLAN_MAX_STAGES) ns = PLAN_MAX_STAGES;
  return ns < 1 ? 1 : (int)ns;
}

int auto_piece_len(int64_t nnz_per_stage) {
  const double want = (double)nnz_per_stage / 7000.0;
  int p = 128;
  while (p * 1.41 < want && p < 4096) p *= 2;
  return p;
}

// Sizes of a device-built plan (upper bounds; see dgnn_plan_layout in the header).
dgnn_plan_layout device_plan_layout(int64_t n, int64_t m, int64_t nnz, int64_t ldb, int ns, int piece) {
  dgnn_plan_layout d;
  d.nstage = ns > 0 ? (ns > PLAN_MAX_STAGES ? PLAN_MAX_STAGES : ns) : auto_stage_count(n, m, nnz, ldb);
  d.piece = piece > 0 ? piece : auto_piece_len(nnz / d.nstage);
  d.hdr_ints = PLAN_HDR_INTS;
  d.ncut_max = nnz / d.piece + 1;                            // per stage
  d.nseg_max = n + nnz / d.piece + 1;                        // per stage
  d.npartial_max = 2 * (nnz / d.piece) + 2 * d.nstage + 2;
  d.ws_ints = PLAN_HDR_INTS + 4 * d.nstage * (d.nseg_max + d.ncut_max);
  return d;
}

unsigned capped_blocks(int64_t work, unsigned cap) {
  const int64_t b = (work + 255) / 256;
  return (unsigned)(b > cap ? cap : (b < 1 ? 1 : b));
}

struct Args {
  const int32_t* rowptr; const int32_t* col; const float* val; int32_t n;
  const float* B; int64_t ldb; float* C; int64_t ldc; int32_t F; float alpha, beta;
  hipStream_t s;
};

template <int VW, int L, int NPASS>
int launch_rows(const Args& a) {
  constexpr int G = 64 / L;
  const int64_t rows_per_block = 4 * G;
  const int64_t gx = (a.n + rows_per_block - 1) / rows_per_block;
  const int W = a.F / VW;
  const int gy = (W + L * NPASS - 1) / (L * NPASS);
  if (gx <= 0 || gx > 0x7fffffffLL || gy > 65535) return fail(DGNN_E_BADARG, "grid out of range%s (%lld)", "", gx);
  ProfScope prof(a.s, K_ROWS);
  hipLaunchKernelGGL((spmm_rows_kernel<VW, L, NPASS>), dim3((unsigned)gx, (unsigned)gy), dim3(256), 0, a.s,
                     a.rowptr, a.col, a.val, a.n, a.B, a.ldb, a.C, a.ldc, a.F, a.alpha, a.beta);
  DGNN_HIP(hipGetLastError());
  return DGNN_OK;
}

template <int VW>
int dispatch_rows(const Args& a) {
  const int W = a.F / VW;   // vector columns
  if (W <= 8) return launch_rows<VW, 8, 1>(a);
  if (W <= 16) return launch_rows<VW, 16, 1>(a);
  if (W <= 32) return launch_rows<VW, 32, 1>(a);
  if (W <= 48) return launch_rows<VW, 16, 3>(a);
  const int passes = (W + 63) / 64;
  if (passes <= 1) return launch_rows<VW, 64, 1>(a);
  if (passes == 2) return launch_rows<VW, 64, 2>(a);
  if (passes == 3) return launch_rows<VW, 64, 3>(a);
  if (passes == 4) return launch_rows<VW, 64, 4>(a);
  if (passes == 5) return launch_rows<VW, 64, 5>(a);
  if (passes == 6) return launch_rows<VW, 64, 6>(a);
  // wider: balance the passes over blockIdx.y chunks of at most 8 passes
  const int chunks = (passes + 7) / 8;
  const int per = (passes + chunks - 1) / chunks;
  switch (per) {
    case 4: return launch_rows<VW, 64, 4>(a);
    case 5: return launch_rows<VW, 64, 5>(a);
    case 6: return launch_rows<VW, 64, 6>(a);
    default: return launch_rows<VW, 64, 8>(a);
  }
}

bool aligned(const void* p, size_t a) { return (reinterpret_cast<uintptr_t>(p) % a) == 0; }

bool overlap(const float* B, int64_t rows_b, int64_t ldb, const float* C, int64_t rows_c, int64_t ldc, int F) {
  if (rows_b <= 0 || rows_c <= 0) return false;
  const uintptr_t b0 = (uintptr_t)B, b1 = b0 + ((uint64_t)(rows_b - 1) * ldb + F) * 4;
  const uintptr_t c0 = (uintptr_t)C, c1 = c0 + ((uint64_t)(rows_c - 1) * ldc + F) * 4;
  return b0 < c1 && c0 < b1;
}

template <int L, int U>
int launch_stage(int vwc, bool wide, const int4* seg, int32_t nseg, const int32_t* col, const float* val,
                 const float* B, int64_t ldb, float* C, int64_t ldc, float* P, int64_t ldp,
                 int32_t F, float alpha, bool fresh, hipStream_t s) {
  constexpr int G = 64 / L;
  int64_t nsegblocks = ((int64_t)nseg + 4 * G - 1) / (4 * G);         // 4 waves x G groups per workgroup
  const int64_t nchunks = (ldb / 4 + L - 1) / L;
  int64_t gcd8 = 8;
  while (nchunks % gcd8) gcd8 >>= 1;
  uint32_t k = g_variant == 3 ? 1u : (uint32_t)(8 / gcd8);             // classes per chunk (see the kernel)
  nsegblocks = (nsegblocks + k - 1) / k * k;                           // equal classes; tail blocks idle
  if (g_variant >= 404 && g_variant < 408) k = 0;
  const int64_t nb = nsegblocks * nchunks;
  if (nb <= 0 || nb > 0x7fffffffLL) return fail(DGNN_E_BADARG, "stage grid out of range%s (%lld)", "", nb);
  ProfScope prof(s, K_STAGE);
#define DGNN_STAGE(VWC, WIDE)                                                                             \
  hipLaunchKernelGGL((spmm_stage_kernel<L, U, VWC, WIDE>), dim3((unsigned)nb), dim3(256), 0, s, seg, nseg, \
                     (uint32_t)nsegblocks, col, val, B, ldb, C, ldc, P, ldp, F, alpha, k, fresh ? 1 : 0)
  if (wide) { if (vwc == 4) DGNN_STAGE(4, true); else if (vwc == 2) DGNN_STAGE(2, true); else DGNN_STAGE(1, true); }
  else { if (vwc == 4) DGNN_STAGE(4, false); else if (vwc == 2) DGNN_STAGE(2, false); else DGNN_STAGE(1, false); }
#undef DGNN_STAGE
  DGNN_HIP(hipGetLastError());
  return DGNN_OK;
}


template <int L, int U>
int launch_hot(int vwc, bool wide, const int4* seg, const int32_t* seg_hot, int32_t nseg, const int32_t* col, const float* val,
               const int32_t* hot_cols, const float* B, int64_t ldb, float* C, int64_t ldc, float* P, int64_t ldp,
               int32_t F, float alpha, bool fresh, hipStream_t s) {
  constexpr int G = 64 / L;
  int64_t nsegblocks = ((int64_t)nseg + HOT_WAVES * G - 1) / (HOT_WAVES * G);
  const int64_t nchunks = (ldb / 4 + L - 1) / L;
  int64_t gcd8 = 8;
  while (nchunks % gcd8) gcd8 >>= 1;
  const uint32_t k = (uint32_t)(8 / gcd8);
  nsegblocks = (nsegblocks + k - 1) / k * k;
  const int64_t nbv = nsegblocks * nchunks;                  // a multiple of 8
  if (nbv <= 0 || nbv > 0x7fffffffLL) return fail(DGNN_E_BADARG, "hot stage grid out of range%s (%lld)", "", nbv);
  static int cus = 0;                                        // one workgroup per CU (its LDS is taken whole)
  if (cus == 0) {
    int dev = 0; hipDeviceProp_t prop;
    DGNN_HIP(hipGetDevice(&dev));
    DGNN_HIP(hipGetDeviceProperties(&prop, dev));
    cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount / 8 * 8 : 256;
    if (cus < 8) cus = 8;
  }
  const unsigned grid = (unsigned)(nbv < cus ? nbv : cus);
  ProfScope prof(s, K_HOT);
#define DGNN_HOT(VWC, WIDE)                                                                                     \
  hipLaunchKernelGGL((spmm_hot_kernel<L, U, VWC, WIDE>), dim3(grid), dim3(HOT_WAVES * 64), 0, s, seg, seg_hot, nseg, \
                     (uint32_t)nsegblocks, (uint32_t)nbv, col, val, hot_cols, B, ldb, C, ldc, P, ldp, F, alpha, k, fresh ? 1 : 0, \
                     g_variant >= 200 && g_variant < 204 ? g_variant - 200 : 0)
  if (wide) { if (vwc == 4) DGNN_HOT(4, true); else if (vwc == 2) DGNN_HOT(2, true); else DGNN_HOT(1, true); }
  else { if (vwc == 4) DGNN_HOT(4, false); else if (vwc == 2) DGNN_HOT(2, false); else DGNN_HOT(1, false); }
#undef DGNN_HOT
  DGNN_HIP(hipGetLastError());
  return DGNN_OK;
}


template <int R, int L, int U>
int launch_bundle(int vwc, bool wide, const int32_t* desc, int32_t ndesc, const int32_t* ucol, const float* uval,
                  const float* B, int64_t ldb, float* C, int64_t ldc, float* P, int64_t ldp,
                  int32_t F, float alpha, bool fresh, hipStream_t s) {
  constexpr int G = 64 / L;
  int64_t nsegblocks = ((int64_t)ndesc + 4 * G - 1) / (4 * G);
  const int64_t nchunks = (ldb / 4 + L - 1) / L;
  int64_t gcd8 = 8;
  while (nchunks % gcd8) gcd8 >>= 1;
  const uint32_t k = (uint32_t)(8 / gcd8);
  nsegblocks = (nsegblocks + k - 1) / k * k;
  const int64_t nb = nsegblocks * nchunks;
  if (nb <= 0 || nb > 0x7fffffffLL) return fail(DGNN_E_BADARG, "bundle grid out of range%s (%lld)", "", nb);
  ProfScope prof(s, K_BUNDLE);
#define DGNN_BUNDLE(VWC, WIDE)                                                                                 \
  hipLaunchKernelGGL((spmm_bundle_kernel<R, L, U, VWC, WIDE>), dim3((unsigned)nb), dim3(256), 0, s, desc, ndesc, \
                     (uint32_t)nsegblocks, ucol, uval, B, ldb, C, ldc, P, ldp, F, alpha, k, fresh ? 1 : 0)
  if (wide) { if (vwc == 4) DGNN_BUNDLE(4, true); else if (vwc == 2) DGNN_BUNDLE(2, true); else DGNN_BUNDLE(1, true); }
  else { if (vwc == 4) DGNN_BUNDLE(4, false); else if (vwc == 2) DGNN_BUNDLE(2, false); else DGNN_BUNDLE(1, false); }
#undef DGNN_BUNDLE
  DGNN_HIP(hipGetLastError());
  return DGNN_OK;
}

// ---------------------------------------------------------------------------------------
// Scratch of the no-allocation entry point (dgnn_spmm_coo_noalloc_f32): row pointer, device-built work plan
// and partial rows of ONE call live in this file-scope array instead of caller workspaces. Every GPU the
// library runs on holds its own copy (at its own address) until the process ends. In host code the array's
// name is only a handle: the device address comes from hipGetSymbolAddress, once per device, and that pointer
// is what kernels, memsets and size checks use.
// ---------------------------------------------------------------------------------------
constexpr size_t NOALLOC_SCRATCH_BYTES = (size_t)DGNN_NOALLOC_SCRATCH_BYTES;
constexpr int NOALLOC_MAX_DEVICES = 64;
}  // namespace (interrupted: a __device__ array wants external or static linkage, not a namespace-scope anonymous one)

static __device__ __align__(256) unsigned char scratch[NOALLOC_SCRATCH_BYTES];

namespace {
void* g_scratch_ptr[NOALLOC_MAX_DEVICES] = {};

// Device address of `scratch` on the current device (looked up on the first call there).
int noalloc_scratch(unsigned char** out) {
  int dev = -1;
  DGNN_HIP(hipGetDevice(&dev));
  if (dev < 0 || dev >= NOALLOC_MAX_DEVICES) return fail(DGNN_E_BADARG, "noalloc: device index out of range%s (%lld)", "", dev);
  void* p = __atomic_load_n(&g_scratch_ptr[dev], __ATOMIC_ACQUIRE);
  if (!p) {
    DGNN_HIP(hipGetSymbolAddress(&p, HIP_SYMBOL(scratch)));
    if (!p) return fail(DGNN_E_HIP, "noalloc: hipGetSymbolAddress returned a null address%s", "");
    __atomic_store_n(&g_scratch_ptr[dev], p, __ATOMIC_RELEASE);      // every thread finds the same address
  }
  *out = static_cast<unsigned char*>(p);
  return DGNN_OK;
}

size_t round256(size_t b) { return (b + 255) / 256 * 256; }
}  // namespace

extern "C" {

int dgnn_version(void) { return 280; }   // 280: dgnn_spmm_strip_overwrite_f32; 270: dgnn_strip_plan.ent lines hold the 32 meta words first, then the 32 values; 260: dgnn_strip_plan, dgnn_spmm_strip_f32, dgnn_prof_end_by_kernel, dgnn_set_variant only in -DDGNN_EXPERIMENTS builds; 250: dgnn_dense_plan, dgnn_spmm_dense_f32; 240: dgnn_bundle_plan, dgnn_spmm_bundle_f32; 230: dgnn_plan.hot_rows / hot_cols / seg_hot; 220: dgnn_spmm_coo_noalloc_f32 (210: dgnn_plan.covers_rows, dgnn_spmm_plan_overwrite_f32)

const char* dgnn_last_error(void) { return g_err; }

int dgnn_prof_begin(void) {
  std::lock_guard<std::mutex> lk(g_prof_mu);
  for (auto& r : g_prof) { (void)hipEventDestroy(r.a); (void)hipEventDestroy(r.b); }
  g_prof.clear();
  g_prof_on.store(true);
  return DGNN_OK;
}

int dgnn_prof_end_by_kernel(double* ms, int64_t* launches, int32_t count) {
  g_prof_on.store(false);
  if (count < 0 || (count > 0 && (!ms || !launches))) return fail(DGNN_E_BADARG, "prof_end_by_kernel: bad argument%s", "");
  std::vector<ProfRec> recs;
  { std::lock_guard<std::mutex> lk(g_prof_mu); recs.swap(g_prof); }
  for (int k = 0; k < count; ++k) { ms[k] = 0; launches[k] = 0; }
  int rc = DGNN_OK;
  for (auto& r : recs) {
    float t = 0.f;
    if (hipEventSynchronize(r.b) != hipSuccess || hipEventElapsedTime(&t, r.a, r.b) != hipSuccess) rc = fail(DGNN_E_HIP, "prof_end: event query failed%s", "");
    else if (r.kind < count) { ms[r.kind] += t; launches[r.kind] += 1; }
    (void)hipEventDestroy(r.a); (void)hipEventDestroy(r.b);
  }
  return rc;
}

int dgnn_prof_end(double* stage_ms, int64_t* stage_launches, double* other_ms, int64_t* other_launches) {
  double ms[K_COUNT]; int64_t cnt[K_COUNT];
  const int rc = dgnn_prof_end_by_kernel(ms, cnt, K_COUNT);
  double a = 0, o = 0; int64_t na = 0, no = 0;
  for (int k = 0; k < K_COUNT; ++k) { if (k < K_MERGE) { a += ms[k]; na += cnt[k]; } else { o += ms[k]; no += cnt[k]; } }
  if (stage_ms) *stage_ms = a;
  if (stage_launches) *stage_launches = na;
  if (other_ms) *other_ms = o;
  if (other_launches) *other_launches = no;
  return rc;
}

#ifdef DGNN_EXPERIMENTS
int dgnn_set_variant(int variant) { g_variant = variant; return DGNN_OK; }
int dgnn_get_variant(void) { return g_variant; }
#endif

int dgnn_coo2csr_i32(const int32_t* row, int64_t nnz, int32_t n, int32_t* rowptr, void* stream) {
  if (n < 0 || nnz < 0 || nnz > 0x7fffffffLL) return fail(DGNN_E_BADARG, "coo2csr: bad sizes%s (nnz=%lld)", "", nnz);
  if (!rowptr || (nnz > 0 && !row)) return fail(DGNN_E_BADARG, "coo2csr: null pointer%s", "");
  hipLaunchKernelGGL(coo2csr_kernel, dim3(capped_blocks(nnz + 1, 4096)), dim3(256), 0, (hipStream_t)stream,
                     row, nnz, n, rowptr);
  DGNN_HIP(hipGetLastError());
  return DGNN_OK;
}

int dgnn_idx_i64_to_i32(const int64_t* src, int64_t count, int32_t* dst, void* stream) {
  if (count < 0) return fail(DGNN_E_BADARG, "idx_i64_to_i32: negative count%s", "");
  if (count == 0) return DGNN_OK;
  if (!src || !dst) return fail(DGNN_E_BADARG, "idx_i64_to_i32: null pointer%s", "");
  hipLaunchKernelGGL(idx_narrow_kernel, dim3(capped_blocks(count, 8192)), dim3(256), 0, (hipStream_t)stream,
                     src, count, dst);
  DGNN_HIP(hipGetLastError());
  return DGNN_OK;
}

int dgnn_pad_rows_f32(const float* src, int64_t lds, float* dst, int64_t ldd, int32_t n, int32_t F, void* stream) {
  if (n < 0 || F < 0) return fail(DGNN_E_BADARG, "pad_rows: negative size%s", "");
  if (n == 0 || ldd == 0) return DGNN_OK;
  if (!src || !dst) return fail(DGNN_E_BADARG, "pad_rows: null pointer%s", "");
  if (lds < F || ldd < F || ldd % 4 != 0 || !aligned(dst, 16))
    return fail(DGNN_E_BADARG, "pad_rows: need lds >= F, ldd >= F, ldd %% 4 == 0, dst 16-byte aligned%s", "");
  if (overlap(src, n, lds, dst, n, ldd, F)) return fail(DGNN_E_BADARG, "pad_rows: src and dst overlap%s", "");
  ProfScope prof((hipStream_t)stream, K_HELPER);
  hipLaunchKernelGGL(pad_rows_kernel, dim3(capped_blocks((int64_t)n * (ldd / 4), 16384)), dim3(256), 0,
                     (hipStream_t)stream, src, lds, dst, ldd, n, F);
  DGNN_HIP(hipGetLastError());
  return DGNN_OK;
}

int dgnn_gather_rows_f32(const float* src, int64_t lds, int32_t nsrc, const int32_t* idx, int32_t k,
                         float* dst, int64_t ldd, int32_t F, void* stream) {
  if (k < 0 || F < 0 || nsrc < 0) return fail(DGNN_E_BADARG, "gather_rows: negative size%s", "");
  if (k == 0 || ldd == 0) return DGNN_OK;
  if (!src || !dst || !idx) return fail(DGNN_E_BADARG, "gather_rows: null pointer%s", "");
  if (nsrc == 0) return fail(DGNN_E_BADARG, "gather_rows: rows requested from an empty source%s", "");
  if (lds < F || ldd < F || ldd % 4 != 0 || !aligned(dst, 16))
    return fail(DGNN_E_BADARG, "gather_rows: need lds >= F, ldd >= F, ldd %% 4 == 0, dst 16-byte aligned%s", "");
  if (overlap(src, nsrc, lds, dst, k, ldd, F)) return fail(DGNN_E_BADARG, "gather_rows: src and dst overlap%s", "");
  const bool src_vec4 = (lds % 4 == 0) && aligned(src, 16);
  ProfScope prof((hipStream_t)stream, K_HELPER);
  hipLaunchKernelGGL(gather_rows_kernel, dim3(capped_blocks((int64_t)k * (ldd / 4), 16384)), dim3(256), 0,
                     (hipStream_t)stream, src, lds, idx, dst, ldd, k, F, src_vec4);
  DGNN_HIP(hipGetLastError());
  return DGNN_OK;
}

int dgnn_spmm_csr_f32(const int32_t* rowptr, const int32_t* col, const float* val,
                      int32_t n, int32_t m, int64_t nnz,
                      const float* B, int64_t ldb, float* C, int64_t ldc, int32_t F,
                      float alpha, float beta, void* stream) {
  if (n < 0 || m < 0 || F < 0 || nnz < 0 || nnz > 0x7fffffffLL)
    return fail(DGNN_E_BADARG, "spmm_csr: bad sizes%s (nnz=%lld)", "", nnz);
  if (n == 0 || F == 0) return DGNN_OK;
  if (!rowptr || !C) return fail(DGNN_E_BADARG, "spmm_csr: null rowptr/C%s", "");
  if (ldc < F) return fail(DGNN_E_BADARG, "spmm_csr: ldc < F%s (ldc=%lld)", "", ldc);
  hipStream_t s = (hipStream_t)stream;
  if (nnz == 0) {
    if (beta == 1.f) return DGNN_OK;
    hipLaunchKernelGGL(scale_rows_kernel, dim3(capped_blocks((int64_t)n * F, 8192)), dim3(256), 0, s, C, ldc, n, F, beta);
    DGNN_HIP(hipGetLastError());
    return DGNN_OK;
  }
  if (!col || !val || !B) return fail(DGNN_E_BADARG, "spmm_csr: null col/val/B%s", "");
  if (ldb < F) return fail(DGNN_E_BADARG, "spmm_csr: ldb < F%s (ldb=%lld)", "", ldb);
  if (overlap(B, m, ldb, C, n, ldc, F)) return fail(DGNN_E_BADARG, "spmm_csr: B and C overlap%s", "");
  Args a{rowptr, col, val, n, B, ldb, C, ldc, F, alpha, beta, s};
  if (F % 4 == 0 && ldb % 4 == 0 && ldc % 4 == 0 && aligned(B, 16) && aligned(C, 16)) return dispatch_rows<4>(a);
  if (F % 2 == 0 && ldb % 2 == 0 && ldc % 2 == 0 && aligned(B, 8) && aligned(C, 8)) return dispatch_rows<2>(a);
  if (!aligned(B, 4) || !aligned(C, 4)) return fail(DGNN_E_BADARG, "spmm_csr: B/C not 4-byte aligned%s", "");
  return dispatch_rows<1>(a);
}

int dgnn_spmm_coo_f32(const int32_t* row, const int32_t* col, const float* val,
                      int32_t n, int32_t m, int64_t nnz, int32_t* rowptr_ws,
                      const float* B, int64_t ldb, float* C, int64_t ldc, int32_t F,
                      float alpha, float beta, void* stream) {
  if (n == 0 || F == 0) return DGNN_OK;
  int rc = dgnn_coo2csr_i32(row, nnz, n, rowptr_ws, stream);
  if (rc != DGNN_OK) return rc;
  return dgnn_spmm_csr_f32(rowptr_ws, col, val, n, m, nnz, B, ldb, C, ldc, F, alpha, beta, stream);
}

int dgnn_plan_layout_for(int32_t n, int32_t m, int64_t nnz, int64_t ldb, int32_t nstage, int32_t piece,
                         dgnn_plan_layout* out) {
  if (n < 0 || m < 0 || nnz < 0 || !out) return fail(DGNN_E_BADARG, "plan_layout_for: bad argument%s", "");
  *out = device_plan_layout(n, m, nnz, ldb, nstage, piece);
  if (out->nseg_max > 0x7fffffffLL) return fail(DGNN_E_BADARG, "plan_layout_for: block too large%s", "");
  return DGNN_OK;
}

int dgnn_plan_build(const int32_t* rowptr, const int32_t* col, int32_t n, int32_t m,
                    const dgnn_plan_layout* layout, int32_t* ws, void* stream) {
  if (!layout || layout->nstage < 1 || layout->nstage > PLAN_MAX_STAGES || layout->piece < 1 || layout->hdr_ints != PLAN_HDR_INTS)
    return fail(DGNN_E_BADARG, "plan_build: layout does not come from dgnn_plan_layout_for%s", "");
  if (n < 0 || m < 0) return fail(DGNN_E_BADARG, "plan_build: negative size%s", "");
  if (!ws || !aligned(ws, 16)) return fail(DGNN_E_BADARG, "plan_build: ws must be 16-byte aligned%s", "");
  hipStream_t s = (hipStream_t)stream;
  const int ns = layout->nstage;
  int* hdr = ws;
  int4* seg = reinterpret_cast<int4*>(ws + PLAN_HDR_INTS);
  int4* cut = seg + ns * layout->nseg_max;
  DGNN_HIP(hipMemsetAsync(hdr, 0, PLAN_HDR_INTS * sizeof(int), s));
  DGNN_HIP(hipMemsetAsync(seg, 0xff, (size_t)(ns * layout->nseg_max) * sizeof(int4), s));   // row = -1: unused slot
  DGNN_HIP(hipMemsetAsync(cut, 0, (size_t)(ns * layout->ncut_max) * sizeof(int4), s));      // count = 0: unused slot
  if (n == 0) return DGNN_OK;
  if (!rowptr) return fail(DGNN_E_BADARG, "plan_build: null pointer%s", "");   // col may be null for an empty block
  const int cbw = (int)(((int64_t)m + ns - 1) / ns) > 0 ? (int)(((int64_t)m + ns - 1) / ns) : 1;
  hipLaunchKernelGGL(plan_hist_kernel, dim3(capped_blocks(n, 1024)), dim3(256), 0, s, rowptr, col, n, ns, cbw, layout->piece, hdr);
  hipLaunchKernelGGL(plan_cursor_kernel, dim3(1), dim3(64), 0, s, hdr, ns);
  hipLaunchKernelGGL(plan_fill_kernel, dim3(capped_blocks(n, 4096)), dim3(256), 0, s, rowptr, col, n, ns, cbw, layout->piece,
                     hdr, seg, layout->nseg_max, cut, layout->ncut_max);
  DGNN_HIP(hipGetLastError());
  return DGNN_OK;
}

int dgnn_coo_aligned_workspace(int32_t n, int32_t m, int64_t nnz, int64_t ldb, int64_t* plan_ws_ints, int64_t* partial_rows) {
  if (n < 0 || m < 0 || nnz < 0 || !plan_ws_ints || !partial_rows) return fail(DGNN_E_BADARG, "coo_aligned_workspace: bad argument%s", "");
  const dgnn_plan_layout d = device_plan_layout(n, m, nnz, ldb, 0, 0);
  *plan_ws_ints = d.ws_ints;
  *partial_rows = d.npartial_max;
  return DGNN_OK;
}

int dgnn_spmm_coo_aligned_f32(const int32_t* row, const int32_t* col, const float* val,
                              int32_t n, int32_t m, int64_t nnz, int32_t* rowptr_ws, int32_t* plan_ws,
                              const float* B, int64_t ldb, float* C, int64_t ldc,
                              float* partial, int64_t ldp, int32_t F,
                              float alpha, float beta, void* stream) {
  if (n == 0 || F == 0) return DGNN_OK;
  if (nnz == 0)                                   // nothing to plan: beta scaling only, as dgnn_spmm_coo_f32 does
    return dgnn_spmm_csr_f32(rowptr_ws, col, val, n, m, 0, B, ldb, C, ldc, F, alpha, beta, stream);
  dgnn_plan_layout d;
  int rc = dgnn_plan_layout_for(n, m, nnz, ldb, 0, 0, &d);
  if (rc != DGNN_OK) return rc;
  rc = dgnn_coo2csr_i32(row, nnz, n, rowptr_ws, stream);
  if (rc != DGNN_OK) return rc;
  rc = dgnn_plan_build(rowptr_ws, col, n, m, &d, plan_ws, stream);
  if (rc != DGNN_OK) return rc;
  int64_t seg_off[PLAN_MAX_STAGES + 1], cut_off[PLAN_MAX_STAGES + 1];
  for (int st = 0; st <= d.nstage; ++st) { seg_off[st] = st * d.nseg_max; cut_off[st] = st * d.ncut_max; }
  dgnn_plan plan;
  plan.nstage = d.nstage; plan.seg = plan_ws + d.hdr_ints; plan.stage_seg_off = seg_off;
  plan.cut = plan_ws + d.hdr_ints + 4 * d.nstage * d.nseg_max; plan.stage_cut_off = cut_off;
  plan.npartial = d.npartial_max; plan.max_seg_len = d.piece; plan.covers_rows = 1;
  plan.hot_rows = 0; plan.hot_cols = nullptr; plan.seg_hot = nullptr;
  return dgnn_spmm_plan_f32(&plan, col, val, n, m, B, ldb, C, ldc, partial, ldp, F, alpha, beta, stream);
}

int dgnn_noalloc_limits(int64_t* scratch_bytes) {
  if (!scratch_bytes) return fail(DGNN_E_BADARG, "noalloc_limits: null pointer%s", "");
  *scratch_bytes = (int64_t)NOALLOC_SCRATCH_BYTES;
  // the set-up of the current device (hipGetSymbolAddress: not a stream operation) happens here too, so that a caller
  // who asks for the limits first -- outside any graph capture -- never pays it inside one
  unsigned char* base = nullptr;
  return noalloc_scratch(&base);
}

int dgnn_noalloc_scratch_needed(int32_t n, int32_t m, int64_t nnz, int64_t ldb, int64_t* bytes) {
  if (n < 0 || m < 0 || nnz < 0 || ldb < 0 || !bytes) return fail(DGNN_E_BADARG, "noalloc_scratch_needed: bad argument%s", "");
  const dgnn_plan_layout d = device_plan_layout(n, m, nnz, ldb, 0, 0);
  *bytes = (int64_t)(round256(((size_t)n + 1) * 4) + round256((size_t)d.ws_ints * 4) + round256((size_t)d.npartial_max * (size_t)ldb * 4));
  return DGNN_OK;
}

int dgnn_spmm_coo_noalloc_f32(const int32_t* row, const int32_t* col, const float* val,
                              int32_t n, int32_t m, int64_t nnz,
                              const float* B, int64_t ldb, float* C, int64_t ldc, int32_t F,
                              float alpha, float beta, void* stream) {
  if (n < 0 || m < 0 || F < 0 || nnz < 0 || nnz > 0x7fffffffLL)
    return fail(DGNN_E_BADARG, "spmm_coo_noalloc: bad sizes%s (nnz=%lld)", "", nnz);
  if (n == 0 || F == 0) return DGNN_OK;
  if (!C || ldc < F || !B || (nnz > 0 && (!row || !col || !val)))
    return fail(DGNN_E_BADARG, "spmm_coo_noalloc: null pointer or ldc < F%s", "");
  if (ldb < F || ldb % 4 != 0 || !aligned(B, 16))
    return fail(DGNN_E_BADARG, "spmm_coo_noalloc: B needs ldb >= F, ldb %% 4 == 0 and a 16-byte aligned base (dgnn_pad_rows_f32)%s", "");
  int64_t need = 0;
  int rc = dgnn_noalloc_scratch_needed(n, m, nnz, ldb, &need);
  if (rc != DGNN_OK) return rc;
  if ((size_t)need > NOALLOC_SCRATCH_BYTES)
    return fail(DGNN_E_BADARG, "spmm_coo_noalloc: this block needs %s%lld bytes of scratch, the library holds "
                "DGNN_NOALLOC_SCRATCH_BYTES (see dgnn_noalloc_limits); use dgnn_spmm_coo_aligned_f32 with caller workspaces", "", need);
  unsigned char* base = nullptr;
  rc = noalloc_scratch(&base);
  if (rc != DGNN_OK) return rc;
  const dgnn_plan_layout d = device_plan_layout(n, m, nnz, ldb, 0, 0);
  int32_t* const rowptr_ws = reinterpret_cast<int32_t*>(base);
  int32_t* const plan_ws = reinterpret_cast<int32_t*>(base + round256(((size_t)n + 1) * 4));
  float* const partial = reinterpret_cast<float*>(base + round256(((size_t)n + 1) * 4) + round256((size_t)d.ws_ints * 4));
  // every word the kernels read is written first by this call (coo2csr fills the row pointer, dgnn_plan_build
  // clears the counters and marks every list slot dead, partial rows are stored before the merge reads them),
  // so what the previous call left in the array does not matter
  return dgnn_spmm_coo_aligned_f32(row, col, val, n, m, nnz, rowptr_ws, plan_ws, B, ldb, C, ldc, partial, ldb, F, alpha, beta, stream);
}

int dgnn_spmm_bundle_f32(const dgnn_bundle_plan* plan, int32_t n, int32_t m,
                         const float* B, int64_t ldb, float* C, int64_t ldc, float* partial, int64_t ldp,
                         int32_t F, float alpha, float beta, void* stream) {
  if (!plan) return fail(DGNN_E_BADARG, "spmm_bundle: null plan%s", "");
  const int R = plan->rows_per_bundle;
  if (R != 2 && R != 4 && R != 6 && R != 8) return fail(DGNN_E_BADARG, "spmm_bundle: rows_per_bundle must be 2, 4, 6 or 8%s", "");
  if (n < 0 || m < 0 || F < 0 || plan->nstage < 0) return fail(DGNN_E_BADARG, "spmm_bundle: negative size%s", "");
  if (n == 0 || F == 0) return DGNN_OK;
  if (!C || ldc < F) return fail(DGNN_E_BADARG, "spmm_bundle: bad C%s", "");
  hipStream_t s = (hipStream_t)stream;
  const bool empty = plan->nstage == 0 || !plan->stage_desc_off || plan->stage_desc_off[plan->nstage] == 0;
  // the first stage lists every row of the block (rows with nothing there in bundles of their own kind), so a
  // product with beta == 0 overwrites C without a zero fill and without reading it
  const bool overwrite = beta == 0.f && !empty && plan->stage_desc_off[1] > plan->stage_desc_off[0];
  if (beta != 1.f && !overwrite) {
    ProfScope prof(s, K_HELPER);
    hipLaunchKernelGGL(scale_rows_kernel, dim3(capped_blocks((int64_t)n * F, 8192)), dim3(256), 0, s, C, ldc, n, F, beta);
    DGNN_HIP(hipGetLastError());
  }
  if (empty) return DGNN_OK;
  if (!plan->desc || !plan->ucol || !plan->uval || !plan->stage_cut_off || !B)
    return fail(DGNN_E_BADARG, "spmm_bundle: null pointer%s", "");
  if (ldb < F || ldb % 4 != 0 || !aligned(B, 16) || !aligned(plan->uval, 16))
    return fail(DGNN_E_BADARG, "spmm_bundle: B needs ldb >= F, ldb %% 4 == 0 and a 16-byte aligned base; uval 16-byte aligned%s", "");
  const int64_t ncut_all = plan->stage_cut_off[plan->nstage];
  if (ncut_all > 0 && (!partial || !plan->cut || ldp < ldb || ldp % 4 != 0 || !aligned(partial, 16)))
    return fail(DGNN_E_BADARG, "spmm_bundle: cut bundles need a 16-byte aligned partial buffer with ldp >= ldb, ldp %% 4 == 0%s", "");
  if (overlap(B, m, ldb, C, n, ldc, F)) return fail(DGNN_E_BADARG, "spmm_bundle: B and C overlap%s", "");
  const int vwc = (ldc % 4 == 0 && aligned(C, 16)) ? 4 : ((ldc % 2 == 0 && aligned(C, 8)) ? 2 : 1);
  const bool wide = ((uint64_t)(m > 0 ? m : 1) * (uint64_t)ldb * 4u) >= 0xffffff00ull;
  const bool wg = ldb >= 256 && g_variant != 16;              // 512-byte slices from 256 floats up, as in the stage kernel
  const int u = (g_variant >= 102 && g_variant <= 108) ? g_variant - 100 : 4;
  bool fresh = overwrite;
  for (int32_t st = 0; st < plan->nstage; ++st) {
    const int64_t d0 = plan->stage_desc_off[st], d1 = plan->stage_desc_off[st + 1];
    if (d1 > d0) {
      const int32_t* desc = plan->desc + d0 * (R + 4);
      const int32_t ndesc = (int32_t)(d1 - d0);
      int rc;
#define DGNN_LB(RR, LL, UU) launch_bundle<RR, LL, UU>(vwc, wide, desc, ndesc, plan->ucol, plan->uval, B, ldb, C, ldc, partial, ldp, F, alpha, fresh, s)
      if (R == 4) rc = wg ? (u == 2 ? DGNN_LB(4, 32, 2) : u == 8 ? DGNN_LB(4, 32, 8) : DGNN_LB(4, 32, 4))
                          : (u == 2 ? DGNN_LB(4, 16, 2) : u == 8 ? DGNN_LB(4, 16, 8) : DGNN_LB(4, 16, 4));
      else if (R == 6) rc = wg ? (u == 2 ? DGNN_LB(6, 32, 2) : DGNN_LB(6, 32, 4)) : (u == 2 ? DGNN_LB(6, 16, 2) : DGNN_LB(6, 16, 4));
      else if (R == 2) rc = wg ? DGNN_LB(2, 32, 4) : DGNN_LB(2, 16, 4);
      else rc = wg ? (u == 2 ? DGNN_LB(8, 32, 2) : DGNN_LB(8, 32, 4)) : (u == 2 ? DGNN_LB(8, 16, 2) : DGNN_LB(8, 16, 4));
#undef DGNN_LB
      if (rc != DGNN_OK) return rc;
    }
    const int64_t p0 = plan->stage_cut_off[st], p1 = plan->stage_cut_off[st + 1];
    if (p1 > p0) {
      ProfScope prof(s, K_HELPER);
      const int32_t nc = (int32_t)(p1 - p0);
      const int4* cutp = reinterpret_cast<const int4*>(plan->cut) + p0;
      launch_merge(vwc, cutp, nc, partial, ldp, C, ldc, F, fresh ? 1 : 0, s);
      DGNN_HIP(hipGetLastError());
    }
    if (d1 > d0 || p1 > p0) fresh = false;
  }
  return DGNN_OK;
}

int dgnn_spmm_dense_f32(const dgnn_dense_plan* plan, int32_t n, int32_t m,
                        const float* B, int64_t ldb, float* C, int64_t ldc, float* partial, int64_t ldp,
                        int32_t F, float alpha, void* stream) {
  if (!plan) return fail(DGNN_E_BADARG, "spmm_dense: null plan%s", "");
  if (plan->rows_per_tile != DENSE_TR) return fail(DGNN_E_BADARG, "spmm_dense: rows_per_tile must be 32%s", "");
  if (n < 0 || m < 0 || F < 0 || plan->npiece < 0 || plan->ncut < 0 || plan->ntile < 0 || plan->nk < 0)
    return fail(DGNN_E_BADARG, "spmm_dense: negative size%s", "");
  if (n == 0 || F == 0 || plan->npiece == 0) return DGNN_OK;
  if (!C || ldc < F) return fail(DGNN_E_BADARG, "spmm_dense: bad C%s", "");
  if (!plan->desc || !plan->rows || !plan->kcol || !plan->kval || !B) return fail(DGNN_E_BADARG, "spmm_dense: null pointer%s", "");
  if (ldb < F || ldb % 4 != 0 || !aligned(B, 16) || !aligned(plan->kval, 4))
    return fail(DGNN_E_BADARG, "spmm_dense: B needs ldb >= F, ldb %% 4 == 0 and a 16-byte aligned base (dgnn_pad_rows_f32)%s", "");
  if (plan->ncut > 0 && (!partial || !plan->cut || ldp < ldb || ldp % 4 != 0 || !aligned(partial, 16)))
    return fail(DGNN_E_BADARG, "spmm_dense: cut tiles need a 16-byte aligned partial buffer with ldp >= ldb, ldp %% 4 == 0%s", "");
  if (overlap(B, m, ldb, C, n, ldc, F)) return fail(DGNN_E_BADARG, "spmm_dense: B and C overlap%s", "");
  if (ldb >= (1ll << 30)) return fail(DGNN_E_BADARG, "spmm_dense: ldb out of range%s", "");
  hipStream_t s = (hipStream_t)stream;
  const int w = ldb <= 64 ? 1 : 2;                           // floats per lane: a 64- or a 128-float chunk per wave
  const int64_t nchunk = (ldb + 64 * w - 1) / (64 * w);
  const int64_t nitems = (int64_t)plan->npiece * nchunk;
  if (nitems > 0x7fffffffLL) return fail(DGNN_E_BADARG, "spmm_dense: grid out of range%s (%lld)", "", nitems);
  const int vw2 = (ldc % 2 == 0 && aligned(C, 8)) ? 1 : 0;
  const int vwc = (ldc % 4 == 0 && aligned(C, 16)) ? 4 : (vw2 ? 2 : 1);
  {
    ProfScope prof(s, K_DENSE);
    const dim3 grid((unsigned)((nitems + 3) / 4));
    const int4* desc = reinterpret_cast<const int4*>(plan->desc);
#define DGNN_DENSE(A1, D, WW)                                                                                          \
    hipLaunchKernelGGL((spmm_dense_kernel<A1, D, WW>), grid, dim3(256), 0, s, desc, (int32_t)nitems, (int32_t)nchunk, plan->rows, \
                       plan->kcol, plan->kval, B, ldb, C, ldc, partial, ldp, F, alpha, vw2)
    if (w == 1) { if (alpha != 1.f) DGNN_DENSE(false, 0, 1); else DGNN_DENSE(true, 0, 1); }
    else if (alpha != 1.f) DGNN_DENSE(false, 0, 2);
#ifdef DGNN_EXPERIMENTS                          // parts of the kernel switched off, for timing (results are wrong)
    else if (g_variant == 301) DGNN_DENSE(true, 1, 2);
    else if (g_variant == 302) DGNN_DENSE(true, 2, 2);
    else if (g_variant == 303) DGNN_DENSE(true, 3, 2);
    else if (g_variant == 307) DGNN_DENSE(true, 7, 2);
    else if (g_variant == 311) DGNN_DENSE(true, 11, 2);
    else if (g_variant == 315) DGNN_DENSE(true, 15, 2);
#endif
    else DGNN_DENSE(true, 0, 2);
#undef DGNN_DENSE
    DGNN_HIP(hipGetLastError());
  }
  if (plan->ncut > 0) {
    ProfScope prof(s, K_MERGE);
    const int4* cutp = reinterpret_cast<const int4*>(plan->cut);
    launch_merge(vwc, cutp, plan->ncut, partial, ldp, C, ldc, F, 0, s);
    DGNN_HIP(hipGetLastError());
  }
  return DGNN_OK;
}

static int spmm_strip_impl(const dgnn_strip_plan* plan, int32_t n, int32_t m,
                           const float* B, int64_t ldb, float* C, int64_t ldc, float* partial, int64_t ldp,
                           int32_t F, float alpha, int fresh, void* stream) {
  if (!plan) return fail(DGNN_E_BADARG, "spmm_strip: null plan%s", "");
  if (plan->rows_per_tile != STRIP_R || plan->strip_cols != STRIP_W || plan->waves != STRIP_NW || plan->line_entries != STRIP_LINE)
    return fail(DGNN_E_BADARG, "spmm_strip: plan geometry must be 256 rows per tile, at most 64 columns per strip, 16 waves, 32 entries per line%s", "");
  if (n < 0 || m < 0 || F < 0 || plan->npiece < 0 || plan->ncut < 0 || plan->ntile < 0)
    return fail(DGNN_E_BADARG, "spmm_strip: negative size%s", "");
  if (n == 0 || F == 0 || plan->npiece == 0) return DGNN_OK;
  if (alpha != 1.f) return fail(DGNN_E_BADARG, "spmm_strip: only alpha == 1 (the values go into the FMAs as stored)%s", "");
  if (!C || ldc < F) return fail(DGNN_E_BADARG, "spmm_strip: bad C%s", "");
  if (!plan->desc || !plan->rows || !plan->ucol || !plan->ent || !plan->ebase || !B)
    return fail(DGNN_E_BADARG, "spmm_strip: null pointer%s", "");
  if (ldb < F || ldb % 32 != 0 || !aligned(B, 16) || !aligned(plan->ent, 8))
    return fail(DGNN_E_BADARG, "spmm_strip: B needs ldb >= F, ldb %% 32 == 0 and a 16-byte aligned base (dgnn_pad_rows_f32 at the padded pitch)%s", "");
  if (plan->ncut > 0 && (!partial || !plan->cut || ldp < ldb || ldp % 4 != 0 || !aligned(partial, 16)))
    return fail(DGNN_E_BADARG, "spmm_strip: cut tiles need a 16-byte aligned partial buffer with ldp >= ldb, ldp %% 4 == 0%s", "");
  if (overlap(B, m, ldb, C, n, ldc, F)) return fail(DGNN_E_BADARG, "spmm_strip: B and C overlap%s", "");
  if (ldb >= (1ll << 30)) return fail(DGNN_E_BADARG, "spmm_strip: ldb out of range%s", "");
  hipStream_t s = (hipStream_t)stream;
  const int64_t nchunk = (ldb + STRIP_CH - 1) / STRIP_CH;
  const int64_t nitems = (int64_t)plan->npiece * nchunk;
  if (nitems > 0x7fffffffLL) return fail(DGNN_E_BADARG, "spmm_strip: grid out of range%s (%lld)", "", nitems);
  const int vw2 = (ldc % 2 == 0 && aligned(C, 8)) ? 1 : 0;
  const int vwc = (ldc % 4 == 0 && aligned(C, 16)) ? 4 : (vw2 ? 2 : 1);
  {
    ProfScope prof(s, K_STRIP);
    // timing experiments (experiments build; wrong sums, in-bounds): variant 500 + bits: 2 no fills, 4 no barrier, 8 no slice reads, 16 no FMAs
    const int dbg = (g_variant >= 500 && g_variant < 532) ? ((g_variant - 500) & 0x1e) : 0;
#define DGNN_STRIP_LAUNCH(LV)                                                                                            \
    hipLaunchKernelGGL(spmm_strip_kernel<LV>, dim3((unsigned)nitems), dim3(STRIP_NW * 64), 0, s, plan->desc, plan->npiece, plan->rows, \
                       plan->ucol, plan->ent, plan->ebase, B, ldb, C, ldc, partial, ldp, F, vw2, dbg & 6, fresh)
#ifdef DGNN_EXPERIMENTS
    if (dbg & 8) DGNN_STRIP_LAUNCH(1);
    else if (dbg & 16) DGNN_STRIP_LAUNCH(2);
    else
#endif
    DGNN_STRIP_LAUNCH(0);
#undef DGNN_STRIP_LAUNCH
    DGNN_HIP(hipGetLastError());
  }
  if (plan->ncut > 0) {
    ProfScope prof(s, K_MERGE);
    launch_merge(vwc, reinterpret_cast<const int4*>(plan->cut), plan->ncut, partial, ldp, C, ldc, F, fresh, s);
    DGNN_HIP(hipGetLastError());
  }
  return DGNN_OK;
}

int dgnn_spmm_strip_f32(const dgnn_strip_plan* plan, int32_t n, int32_t m,
                        const float* B, int64_t ldb, float* C, int64_t ldc, float* partial, int64_t ldp,
                        int32_t F, float alpha, void* stream) {
  return spmm_strip_impl(plan, n, m, B, ldb, C, ldc, partial, ldp, F, alpha, 0, stream);
}

int dgnn_spmm_strip_overwrite_f32(const dgnn_strip_plan* plan, int32_t n, int32_t m,
                                  const float* B, int64_t ldb, float* C, int64_t ldc, float* partial, int64_t ldp,
                                  int32_t F, float alpha, void* stream) {
  return spmm_strip_impl(plan, n, m, B, ldb, C, ldc, partial, ldp, F, alpha, 1, stream);
}

static int spmm_plan_impl(const dgnn_plan* plan, const int32_t* col, const float* val,
                          int32_t n, int32_t m, const float* B, int64_t ldb,
                          float* C, int64_t ldc, float* partial, int64_t ldp,
                          int32_t F, float alpha, float beta, void* stream) {
  if (!plan) return fail(DGNN_E_BADARG, "spmm_plan: null plan%s", "");
  if (n < 0 || m < 0 || F < 0) return fail(DGNN_E_BADARG, "spmm_plan: negative size%s", "");
  if (n == 0 || F == 0) return DGNN_OK;
  if (!C || ldc < F) return fail(DGNN_E_BADARG, "spmm_plan: bad C%s", "");
  hipStream_t s = (hipStream_t)stream;
  // Stages accumulate into C one after another and a row need not appear in stage 0, so C holds
  // beta*C before the first stage.
  const bool empty = plan->nstage <= 0 || !plan->stage_seg_off || plan->stage_seg_off[plan->nstage] == 0;
  // beta == 0 with a plan whose first stage lists every row: nothing to prepare, C is overwritten
  const bool overwrite = beta == 0.f && !empty && plan->covers_rows != 0 && plan->stage_seg_off[1] > plan->stage_seg_off[0];
  if (beta != 1.f && !overwrite) {
    ProfScope prof(s, K_HELPER);
    hipLaunchKernelGGL(scale_rows_kernel, dim3(capped_blocks((int64_t)n * F, 8192)), dim3(256), 0, s, C, ldc, n, F, beta);
    DGNN_HIP(hipGetLastError());
  }
  if (empty) return DGNN_OK;
  if (plan->max_seg_len <= 0)
    return fail(DGNN_E_BADARG, "spmm_plan: plan->max_seg_len must be positive%s (%lld)", "", plan->max_seg_len);
  if (!plan->seg || !plan->stage_cut_off || !col || !val || !B)
    return fail(DGNN_E_BADARG, "spmm_plan: null pointer%s", "");
  if (ldb < F || ldb % 4 != 0 || !aligned(B, 16))
    return fail(DGNN_E_BADARG, "spmm_plan: B needs ldb >= F, ldb %% 4 == 0 and a 16-byte aligned base (dgnn_pad_rows_f32)%s", "");
  if (overlap(B, m, ldb, C, n, ldc, F)) return fail(DGNN_E_BADARG, "spmm_plan: B and C overlap%s", "");
  const int64_t ncut = plan->stage_cut_off[plan->nstage];
  if (ncut > 0 && (!partial || !plan->cut || ldp < ldb || ldp % 4 != 0 || !aligned(partial, 16)))
    return fail(DGNN_E_BADARG, "spmm_plan: cut segments need a 16-byte aligned partial buffer with ldp >= ldb, ldp %% 4 == 0%s", "");
  const int vwc = (ldc % 4 == 0 && aligned(C, 16)) ? 4 : ((ldc % 2 == 0 && aligned(C, 8)) ? 2 : 1);
  bool fresh = beta == 0.f;                    // C holds nothing (or zeros) until the first stage and its merge have run
  const bool wide = ((uint64_t)(m > 0 ? m : 1) * (uint64_t)ldb * 4u) >= 0xffffff00ull || g_variant == 2;   // 32-bit row offsets?
  for (int32_t st = 0; st < plan->nstage; ++st) {
    const int64_t s0 = plan->stage_seg_off[st], s1 = plan->stage_seg_off[st + 1];
    if (s1 > s0) {
      const int4* seg = reinterpret_cast<const int4*>(plan->seg) + s0;
      int rc;
      const int32_t ns = (int32_t)(s1 - s0);
      // 16-lane groups (256-B slices of a B row per gather), 32-lane groups for wide blocks. 8-lane groups (128-B slices) looked
      // better for narrow blocks only while XCD shares were unbalanced (more chunks hid it); with balanced
      // shares they are 5-25 % slower at every width measured. Kept behind the variant knob for A/B runs.
      const bool narrow = (g_variant == 8);
      const int lsel = (g_variant >= 400 && g_variant < 408) ? (g_variant & 3) : 0;     // experiments: 1 -> 8, 2 -> 16, 3 -> 32 lanes
      if (plan->hot_rows > 0) {
        if ((plan->hot_rows != 256 && plan->hot_rows != 512) || !plan->hot_cols || !plan->seg_hot)
          return fail(DGNN_E_BADARG, "spmm_plan: hot-tile plan needs hot_rows 256 or 512, hot_cols and seg_hot%s", "");
        const int32_t* hc = plan->hot_cols + (int64_t)st * plan->hot_rows;
        const int32_t* sh = plan->seg_hot + s0;
        const int u = (g_variant >= 102 && g_variant <= 108) ? g_variant - 100 : 4;
#define DGNN_LAUNCH_HOT(LL, UU) launch_hot<LL, UU>(vwc, wide, seg, sh, ns, col, val, hc, B, ldb, C, ldc, partial, ldp, F, alpha, fresh, s)
        if (plan->hot_rows == 512) rc = (u == 2) ? DGNN_LAUNCH_HOT(16, 2) : (u == 8) ? DGNN_LAUNCH_HOT(16, 8) : DGNN_LAUNCH_HOT(16, 4);
        else rc = (u == 2) ? DGNN_LAUNCH_HOT(32, 2) : (u == 8) ? DGNN_LAUNCH_HOT(32, 8) : DGNN_LAUNCH_HOT(32, 4);
#undef DGNN_LAUNCH_HOT
      } else if (narrow)
        rc = launch_stage<8, 8>(vwc, wide, seg, ns, col, val, B, ldb, C, ldc, partial, ldp, F, alpha, fresh, s);
      else if (lsel == 1)
        rc = launch_stage<8, 2>(vwc, wide, seg, ns, col, val, B, ldb, C, ldc, partial, ldp, F, alpha, fresh, s);
      else {
        // Gathers in flight per lane: 4 while one stage's slice of B fits the 256 MiB Infinity Cache, 2 when
        // the rows come out of HBM (measured: products/papers shapes -6..-10 % with 2, Reddit -4 % with 4
        // against the 8 of earlier rounds -- more outstanding random reads only queue up).
        const int64_t stage_bytes = (int64_t)(m / plan->nstage + 1) * ldb * 4;
        // A plan that leaves very long runs whole (the plan-less reference call) is bound by the serial walk
        // of its longest run: there, 8 in flight.
        const bool wide_group = lsel ? lsel == 3 : (g_variant == 32 || (g_variant != 16 && ldb >= 256));   // 512-B slices from 256 floats up
        // ... and 8 again when the launch is small (under ~200 K lane groups: the blocks of a 4- or 8-rank
        // partition, or one-chunk-wide features): few waves per CU, so each must keep more in flight and
        // the longest pieces set the time (measured -2..-25 %).
        const int64_t lane_groups = (int64_t)ns * ((ldb / 4 + (wide_group ? 32 : 16) - 1) / (wide_group ? 32 : 16));
        const int u = (g_variant >= 102 && g_variant <= 108) ? g_variant - 100
                      : (plan->max_seg_len > 16384 || lane_groups < 200000) ? 8
                      : (stage_bytes <= (256ll << 20) ? 4 : 2);
#define DGNN_LAUNCH(LL, UU) launch_stage<LL, UU>(vwc, wide, seg, ns, col, val, B, ldb, C, ldc, partial, ldp, F, alpha, fresh, s)
        if (wide_group) rc = (u == 2) ? DGNN_LAUNCH(32, 2) : (u == 4) ? DGNN_LAUNCH(32, 4) : (u == 6) ? DGNN_LAUNCH(32, 6) : DGNN_LAUNCH(32, 8);
        else rc = (u == 2) ? DGNN_LAUNCH(16, 2) : (u == 4) ? DGNN_LAUNCH(16, 4) : (u == 6) ? DGNN_LAUNCH(16, 6) : DGNN_LAUNCH(16, 8);
#undef DGNN_LAUNCH
      }
      if (rc != DGNN_OK) return rc;
    }
    const int64_t p0 = plan->stage_cut_off[st], p1 = plan->stage_cut_off[st + 1];
    if (p1 > p0) {
      ProfScope prof(s, K_HELPER);
      const int32_t nc = (int32_t)(p1 - p0);
      const int4* cutp = reinterpret_cast<const int4*>(plan->cut) + p0;
      launch_merge(vwc, cutp, nc, partial, ldp, C, ldc, F, fresh ? 1 : 0, s);
      DGNN_HIP(hipGetLastError());
    }
    if (s1 > s0 || p1 > p0) fresh = false;
  }
  return DGNN_OK;
}

int dgnn_spmm_plan_f32(const dgnn_plan* plan, const int32_t* col, const float* val,
                       int32_t n, int32_t m, const float* B, int64_t ldb,
                       float* C, int64_t ldc, float* partial, int64_t ldp,
                       int32_t F, float alpha, float beta, void* stream) {
  return spmm_plan_impl(plan, col, val, n, m, B, ldb, C, ldc, partial, ldp, F, alpha, beta, stream);
}

int dgnn_spmm_plan_overwrite_f32(const dgnn_plan* plan, const int32_t* col, const float* val,
                                 int32_t n, int32_t m, const float* B, int64_t ldb,
                                 float* C, int64_t ldc, float* partial, int64_t ldp,
                                 int32_t F, float alpha, void* stream) {
  return spmm_plan_impl(plan, col, val, n, m, B, ldb, C, ldc, partial, ldp, F, alpha, 0.f, stream);
}

}  // extern "C"
