// Probe: does gfx950 honour VGPR index mode (s_set_gpr_idx_*) on v_fma_f32 / v_pk_fma_f32 dst + src2?
// acc[r] = fma(v, b, acc[r]) with a wave-uniform dynamic r, as ONE VALU instruction.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
typedef float f32x32 __attribute__((ext_vector_type(32)));

__global__ void probe(const int* __restrict__ idx, const float* __restrict__ vals, float* __restrict__ out, int n, int mode) {
  f32x32 acc;
  for (int j = 0; j < 32; ++j) acc[j] = 0.f;
  const float b = (float)(threadIdx.x + 1);
  for (int i = 0; i < n; ++i) {
    const int r = __builtin_amdgcn_readfirstlane(idx[i]);
    const float v = __builtin_amdgcn_readfirstlane(__float_as_int(vals[i])) ? __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(vals[i]))) : 0.f;
    if (mode == 0) {
      asm volatile("s_set_gpr_idx_on %1, 0xc\n\t"
                   "v_fma_f32 v64, %2, %3, v64\n\t"
                   "s_set_gpr_idx_off"
                   : "+{v[64:95]}"(acc) : "s"(r), "s"(v), "v"(b));
    } else {
      asm volatile("s_set_gpr_idx_on %1, 0xc\n\t"
                   "v_pk_fma_f32 v[64:65], %2, %3, v[64:65]\n\t"
                   "s_set_gpr_idx_off"
                   : "+{v[64:95]}"(acc) : "s"(r & ~1), "v"((float2){v, v}), "v"((float2){b, b + 0.5f}));
    }
  }
  for (int j = 0; j < 32; ++j) out[j * 64 + threadIdx.x] = acc[j];
}

int main() {
  const int n = 200;
  std::vector<int> idx(n); std::vector<float> vals(n);
  for (int i = 0; i < n; ++i) { idx[i] = (i * 7 + 3) % 32; vals[i] = 0.25f * (i % 9 + 1); }
  int* d_idx; float *d_vals, *d_out;
  hipMalloc(&d_idx, n * 4); hipMalloc(&d_vals, n * 4); hipMalloc(&d_out, 32 * 64 * 4);
  hipMemcpy(d_idx, idx.data(), n * 4, hipMemcpyHostToDevice);
  hipMemcpy(d_vals, vals.data(), n * 4, hipMemcpyHostToDevice);
  for (int mode = 0; mode < 2; ++mode) {
    hipMemset(d_out, 0, 32 * 64 * 4);
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d_idx, d_vals, d_out, n, mode);
    hipError_t e = hipDeviceSynchronize();
    if (e != hipSuccess) { printf("mode %d: %s\n", mode, hipGetErrorString(e)); return 1; }
    std::vector<float> out(32 * 64);
    hipMemcpy(out.data(), d_out, 32 * 64 * 4, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int lane = 0; lane < 64; ++lane) {
      float want[32] = {0};
      for (int i = 0; i < n; ++i) {
        if (mode == 0) want[idx[i]] = fmaf(vals[i], (float)(lane + 1), want[idx[i]]);
        else { int r = idx[i] & ~1; want[r] = fmaf(vals[i], (float)(lane + 1), want[r]); want[r + 1] = fmaf(vals[i], (float)(lane + 1) + 0.5f, want[r + 1]); }
      }
      for (int j = 0; j < 32; ++j) if (out[j * 64 + lane] != want[j]) { if (bad < 5) printf("mode %d lane %d acc[%d] = %g want %g\n", mode, lane, j, out[j * 64 + lane], want[j]); ++bad; }
    }
    printf("mode %d (%s): %s (%d mismatches)\n", mode, mode ? "v_pk_fma_f32" : "v_fma_f32", bad ? "FAIL" : "ok", bad);
  }
  return 0;
}
