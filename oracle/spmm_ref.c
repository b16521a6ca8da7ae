/*
 * oracle/spmm_ref.c -- TEST INFRASTRUCTURE ONLY (the CPU oracle). Nothing in the product
 * path (dist-gnn_amd/) may link, load or call this file; only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg do, as the checker / the timed CPU baseline.
 *
 * Plain-C restatement of the arithmetic the reference's SpMM operator performs:
 *
 *   C = alpha * A * B + beta * C,   A sparse COO (row-sorted, coalesced), B/C dense row-major
 *
 * Reference call sites:  light_dist_gnn/models/gcn.py:11 (spmm_cusparse(..., 1, 1)) and its
 * documented fallback gcn.py:13 `C.addmm_(A, B)`; native op light_dist_gnn/spmm_cpp/spmm.cpp:17-33.
 * The GPU library behind spmm.cpp (cuSPARSE, closed source, CUDA 10.2/11.1 per
 * light_dist_gnn/readme.md) is absent from /root/reference and cannot be built on this ROCm image
 * (needs cusparse.h): "unbuildable" -- there is no oracle/_ref. The fallback lives in a third-party
 * dependency as well: ATen's sparse-COO x dense addmm on CPU (torch; the reference pins no
 * version beyond "pytorch-lts"; this image has torch 2.10.0), whose published algorithm
 * (aten/src/ATen/native/sparse/SparseTensorMath.cpp, s_addmm_out_sparse_dense_worker) is:
 *
 *     r = beta * t   (r.zero_() when beta == 0)
 *     for k in 0 .. nnz-1 in coalesced (row-major sorted) order:
 *         axpy(F, alpha * val[k], B[col[k], :], r[row[k], :])
 *
 * i.e. one sequential accumulation chain per output element, in ascending nonzero order.
 * That is what this file restates. Parity is pinned by tests/golden/ (vectors produced by
 * running the reference's own Python -- graph_utils.py, models/gcn.py with the addmm_ fallback --
 * in the build container; generator script tests/golden/make_golden.py).
 *
 * axpy is restated with fmaf(): BLAS saxpy kernels on x86-64 with FMA fuse the multiply-add,
 * and the HIP kernel uses v_fma_f32 too, so the oracle and the GPU agree bit for bit whenever
 * the GPU keeps the ascending order (it does for non-split rows).
 */
#include <math.h>
#include <stdint.h>
#include <string.h>

/* sorted COO row indices -> CSR row pointer (cusparseXcoo2csr, spmm_original.cpp:141-146) */
void oracle_coo2csr_i32(const int32_t* row, int64_t nnz, int32_t n, int32_t* rowptr) {
  int64_t k = 0;
  for (int32_t r = 0; r <= n; ++r) {
    while (k < nnz && row[k] < r) ++k;
    rowptr[r] = (int32_t)k;
  }
}

/* ATen s_addmm_out_sparse_dense_worker restated; nonzeros visited in storage order. */
void oracle_spmm_coo_f32(const int32_t* row, const int32_t* col, const float* val, int64_t nnz,
                         int32_t n, int32_t m, const float* B, int64_t ldb, float* C, int64_t ldc,
                         int32_t F, float alpha, float beta) {
  (void)m;
  for (int32_t i = 0; i < n; ++i) {
    float* c = C + (int64_t)i * ldc;
    if (beta == 0.0f) memset(c, 0, sizeof(float) * (size_t)F);
    else if (beta != 1.0f) for (int32_t f = 0; f < F; ++f) c[f] = beta * c[f];
  }
  for (int64_t k = 0; k < nnz; ++k) {
    const float a = alpha * val[k];
    const float* b = B + (int64_t)col[k] * ldb;
    float* c = C + (int64_t)row[k] * ldc;
    for (int32_t f = 0; f < F; ++f) c[f] = fmaf(a, b[f], c[f]);
  }
}

/* Same chain driven from CSR; rows [r0, r1) only so callers can time a bounded sample. */
void oracle_spmm_csr_f32(const int32_t* rowptr, const int32_t* col, const float* val,
                         int32_t r0, int32_t r1, const float* B, int64_t ldb, float* C, int64_t ldc,
                         int32_t F, float alpha, float beta) {
  for (int32_t i = r0; i < r1; ++i) {
    float* c = C + (int64_t)i * ldc;
    if (beta == 0.0f) memset(c, 0, sizeof(float) * (size_t)F);
    else if (beta != 1.0f) for (int32_t f = 0; f < F; ++f) c[f] = beta * c[f];
    for (int32_t k = rowptr[i]; k < rowptr[i + 1]; ++k) {
      const float a = alpha * val[k];
      const float* b = B + (int64_t)col[k] * ldb;
      for (int32_t f = 0; f < F; ++f) c[f] = fmaf(a, b[f], c[f]);
    }
  }
}

/* The same per-row chain with the rows dealt to `threads` host threads (OpenMP). Rows are independent,
 * so the result is bit-identical to oracle_spmm_csr_f32; this is the multi-core CPU baseline that
 * bench.py times (the reference's CPU path is ATen's addmm_ under torch.set_num_threads). */
void oracle_spmm_csr_f32_mt(const int32_t* rowptr, const int32_t* col, const float* val,
                            int32_t r0, int32_t r1, const float* B, int64_t ldb, float* C, int64_t ldc,
                            int32_t F, float alpha, float beta, int32_t threads) {
  if (threads < 1) threads = 1;
#pragma omp parallel for schedule(dynamic, 16) num_threads(threads)
  for (int32_t i = r0; i < r1; ++i)
    oracle_spmm_csr_f32(rowptr, col, val, i, i + 1, B, ldb, C, ldc, F, alpha, beta);
}

/* fp64 accumulation of the same sum and of sum|a*b| -- gives the tests an error yardstick:
 * |fp32 result - exact| <= gamma_n * abs_sum. */
void oracle_spmm_csr_f64(const int32_t* rowptr, const int32_t* col, const float* val,
                         int32_t n, const float* B, int64_t ldb, const float* Cin, int64_t ldc,
                         int32_t F, float alpha, float beta, double* out, double* abs_out) {
  for (int32_t i = 0; i < n; ++i) {
    double* o = out + (int64_t)i * F;
    double* ab = abs_out + (int64_t)i * F;
    for (int32_t f = 0; f < F; ++f) {
      const double c0 = (beta == 0.0f) ? 0.0 : (double)beta * (double)Cin[(int64_t)i * ldc + f];
      o[f] = c0; ab[f] = fabs(c0);
    }
    for (int32_t k = rowptr[i]; k < rowptr[i + 1]; ++k) {
      const double a = (double)alpha * (double)val[k];
      const float* b = B + (int64_t)col[k] * ldb;
      for (int32_t f = 0; f < F; ++f) { o[f] += a * (double)b[f]; ab[f] += fabs(a * (double)b[f]); }
    }
  }
}
