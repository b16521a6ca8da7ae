// dgnn_spmm.hip -- hand-written gfx950 (MI355X / CDNA4) kernels behind the C ABI in
// include/dgnn_spmm.h. Replaces dist-gnn's cuSPARSE wrapper
// (reference: light_dist_gnn/spmm_cpp/spmm.cpp:17-33, spmm_original.cpp:114-193).
//
// The SpMM is a gather: every nonzero (i, j, a) pulls the F floats of B[j, :] through the
// cache hierarchy and fuses them into the accumulators of C[i, :]. There is no MFMA here;
// the limits are the vector-memory (TA/L1 -> L2 -> Infinity Cache -> HBM) gather rate and,
// for the algorithmic byte count, HBM. Layout and mapping:
//
//   * CSR (int32 rowptr/col, fp32 val) -- (col, val) of a row are read as coalesced
//     64-entry (or L-entry) segments, one entry per lane, and handed lane-to-lane with
//     v_readlane (wave-uniform column -> scalar row base) or ds_bpermute (sub-wave groups).
//   * A "group" of L lanes (L = 64, 32, 16, 8) owns one output row; lane li owns VW
//     consecutive floats (VW = 4/2/1 -> global_load_dwordx4/x2/x1) of each of NPASS
//     feature passes, so every output element has exactly one accumulator that sees the
//     row's nonzeros in ascending order: bit-reproducible, no atomics.
//   * Feature chunks wider than one group pass go to blockIdx.y.
//
// 64-wide wavefronts are assumed throughout (gfx950 only).

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include "dgnn_spmm.h"

namespace {

thread_local char g_err[512] = "";
int g_variant = 0;

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

// Hand one (col, val) entry from lane `src` of the group to every lane of the group.
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
// Row-per-group CSR SpMM.  grid.x = ceil(n / (4 waves * G rows)), grid.y = feature chunks.
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

  // acc starts from beta*C and every nonzero adds fma(alpha*a, b, acc) in ascending order:
  // the same chain as ATen's CPU sparse addmm (oracle/spmm_ref.c), so results agree bit for bit.
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
// coo2csr: thread k closes every row boundary between row[k-1] and row[k].
// ---------------------------------------------------------------------------------------
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

// beta-only path for an empty A (nnz == 0): C = beta * C.
__global__ __launch_bounds__(256) void scale_rows_kernel(float* __restrict__ C, int64_t ldc, int32_t n,
                                                         int32_t F, float beta) {
  const int64_t total = (int64_t)n * F;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += stride) {
    float* p = C + (t / F) * ldc + (t % F);
    *p = (beta == 0.f) ? 0.f : beta * *p;
  }
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

}  // namespace

extern "C" {

int dgnn_version(void) { return 100; }

const char* dgnn_last_error(void) { return g_err; }

int dgnn_set_variant(int variant) { g_variant = variant; return DGNN_OK; }
int dgnn_get_variant(void) { return g_variant; }

int dgnn_coo2csr_i32(const int32_t* row, int64_t nnz, int32_t n, int32_t* rowptr, void* stream) {
  if (n < 0 || nnz < 0 || nnz > 0x7fffffffLL) return fail(DGNN_E_BADARG, "coo2csr: bad sizes%s (nnz=%lld)", "", nnz);
  if (!rowptr || (nnz > 0 && !row)) return fail(DGNN_E_BADARG, "coo2csr: null pointer%s", "");
  const int64_t work = nnz + 1;
  const unsigned blocks = (unsigned)((work + 255) / 256 > 4096 ? 4096 : (work + 255) / 256);
  hipLaunchKernelGGL(coo2csr_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, row, nnz, n, rowptr);
  DGNN_HIP(hipGetLastError());
  return DGNN_OK;
}

int dgnn_idx_i64_to_i32(const int64_t* src, int64_t count, int32_t* dst, void* stream) {
  if (count < 0) return fail(DGNN_E_BADARG, "idx_i64_to_i32: negative count%s", "");
  if (count == 0) return DGNN_OK;
  if (!src || !dst) return fail(DGNN_E_BADARG, "idx_i64_to_i32: null pointer%s", "");
  const unsigned blocks = (unsigned)((count + 255) / 256 > 8192 ? 8192 : (count + 255) / 256);
  hipLaunchKernelGGL(idx_narrow_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, src, count, dst);
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
    const int64_t total = (int64_t)n * F;
    const unsigned blocks = (unsigned)((total + 255) / 256 > 8192 ? 8192 : (total + 255) / 256);
    hipLaunchKernelGGL(scale_rows_kernel, dim3(blocks), dim3(256), 0, s, C, ldc, n, F, beta);
    DGNN_HIP(hipGetLastError());
    return DGNN_OK;
  }
  if (!col || !val || !B) return fail(DGNN_E_BADARG, "spmm_csr: null col/val/B%s", "");
  if (ldb < F) return fail(DGNN_E_BADARG, "spmm_csr: ldb < F%s (ldb=%lld)", "", ldb);
  {  // B and C must not overlap (the kernel reads B rows while other waves write C rows)
    const uintptr_t b0 = (uintptr_t)B, b1 = b0 + ((uint64_t)(m > 0 ? m - 1 : 0) * ldb + F) * 4;
    const uintptr_t c0 = (uintptr_t)C, c1 = c0 + ((uint64_t)(n - 1) * ldc + F) * 4;
    if (m > 0 && b0 < c1 && c0 < b1) return fail(DGNN_E_BADARG, "spmm_csr: B and C overlap%s", "");
  }
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

}  // extern "C"
