// Micro-benchmark: what fp32 FMA rate do v_pk_fma_f32 streams reach on this GPU (register operands only), and at what
// shader clock? Decides what "FMA-bound" means for spmm_dense_kernel.   hipcc --offload-arch=gfx950 -O3 fma_peak.hip -o fma_peak
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef float f32x2 __attribute__((ext_vector_type(2)));

template <bool SGPR>
__global__ __launch_bounds__(256) void fma_kernel(float* out, const float* __restrict__ a, int iters, long long* clk) {
  f32x2 acc[32];
#pragma unroll
  for (int j = 0; j < 32; ++j) acc[j] = (f32x2){(float)j, (float)threadIdx.x};
  const long long t0 = clock64(), w0 = wall_clock64();
  f32x2 b = {out[threadIdx.x & 7], out[(threadIdx.x + 1) & 7]};
  for (int it = 0; it < iters; ++it) {
    const float* ap = a + (it & 15) * 64;
#pragma unroll
    for (int j = 0; j < 32; ++j) {
      f32x2 a2;
      if (SGPR) a2 = (f32x2){ap[2 * j], ap[2 * j + 1]};
      else a2 = (f32x2){b.y, b.x};
      acc[j] = __builtin_elementwise_fma(a2, b, acc[j]);
    }
  }
  const long long t1 = clock64(), w1 = wall_clock64();
  float s = 0.f;
#pragma unroll
  for (int j = 0; j < 32; ++j) s += acc[j].x + acc[j].y;
  out[8 + blockIdx.x * 256 + threadIdx.x] = s;
  if (blockIdx.x == 0 && threadIdx.x == 0) { clk[0] = t1 - t0; clk[1] = w1 - w0; }
}

int main() {
  float *out, *a; long long* clk;
  const int blocks = 256 * 8;
  hipMalloc(&out, (8 + blocks * 256) * sizeof(float)); hipMalloc(&a, 16 * 64 * sizeof(float)); hipMalloc(&clk, 16);
  hipMemset(out, 0, (8 + blocks * 256) * sizeof(float)); hipMemset(a, 0, 16 * 64 * sizeof(float));
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int lds_opts[3] = {0, 40 << 10, 80 << 10};             // dynamic LDS per block: all / 4 / 2 blocks (waves per SIMD) per CU
  for (int lo = 0; lo < 3; ++lo)
  for (int mode = 0; mode < 2; ++mode) {
    const int lds = lds_opts[lo];
    if (lds) { hipFuncSetAttribute((const void*)fma_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, lds); hipFuncSetAttribute((const void*)fma_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, lds); }
    printf("dynamic LDS %d KiB per block:\n", lds >> 10);
    for (int rep = 0; rep < 2; ++rep) {
      const int iters = 20000;
      hipEventRecord(e0);
      if (mode) hipLaunchKernelGGL(fma_kernel<true>, dim3(blocks), dim3(256), lds, 0, out, a, iters, clk);
      else hipLaunchKernelGGL(fma_kernel<false>, dim3(blocks), dim3(256), lds, 0, out, a, iters, clk);
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      long long h[2]; hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost);
      const double fma = (double)blocks * 256 * iters * 32 * 2;
      printf("%s operand: %.3f ms, %.1f T FMA/s = %.1f TFLOP/s; clock64/wall_clock64 = %.3f (x100 MHz => %.0f MHz)\n", mode ? "SGPR" : "VGPR",
             ms, fma / ms / 1e9, 2 * fma / ms / 1e9, (double)h[0] / h[1], 100.0 * h[0] / h[1]);
    }
  }
  return 0;
}
