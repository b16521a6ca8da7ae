// Probe of the strip kernel's inner loop (tools/micro; not part of the library): wave-uniform (slot, row, val) entries
// through scalar loads, B slices from LDS (ds_read_b64), accumulator row selected by VGPR index mode.
// Measures entries per second and checks one workgroup against the host.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <math.h>
#include <string.h>
#include <vector>
typedef float f32x32 __attribute__((ext_vector_type(32)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

#define W 64
#define NW 8
#define LDS_BYTES ((W / 2 + 1) * 2048)

#define PROC1(X, K, B, A)                                                                                   \
  "s_set_gpr_idx_idx s" #X "\n\t"                                                                             \
  "s_waitcnt lgkmcnt(" #K ")\n\t"                                                                            \
  "v_pk_fma_f32 v[64:65], s[" #X ":" #A "], %[" #B "], v[64:65] op_sel:[1,0,0] op_sel_hi:[1,1,1]\n\t"

// X0..X15 = the 16 SGPRs of a batch
#define PROCESS(X0, X1, X2, X3, X4, X5, X6, X7, X8, X9, X10, X11, X12, X13, X14, X15)                       \
  "v_and_or_b32 %[a0], s" #X0 ", %[mask], %[lb]\n\t"                                                        \
  "v_and_or_b32 %[a1], s" #X2 ", %[mask], %[lb]\n\t"                                                        \
  "v_and_or_b32 %[a2], s" #X4 ", %[mask], %[lb]\n\t"                                                        \
  "v_and_or_b32 %[a3], s" #X6 ", %[mask], %[lb]\n\t"                                                        \
  "v_and_or_b32 %[a4], s" #X8 ", %[mask], %[lb]\n\t"                                                        \
  "v_and_or_b32 %[a5], s" #X10 ", %[mask], %[lb]\n\t"                                                       \
  "v_and_or_b32 %[a6], s" #X12 ", %[mask], %[lb]\n\t"                                                       \
  "v_and_or_b32 %[a7], s" #X14 ", %[mask], %[lb]\n\t"                                                       \
  "ds_read_b64 %[b0], %[a0]\n\t"                                                                            \
  "ds_read_b64 %[b1], %[a1]\n\t"                                                                            \
  "ds_read_b64 %[b2], %[a2]\n\t"                                                                            \
  "ds_read_b64 %[b3], %[a3]\n\t"                                                                            \
  "ds_read_b64 %[b4], %[a4]\n\t"                                                                            \
  "ds_read_b64 %[b5], %[a5]\n\t"                                                                            \
  "ds_read_b64 %[b6], %[a6]\n\t"                                                                            \
  "ds_read_b64 %[b7], %[a7]\n\t"                                                                            \
  "s_set_gpr_idx_on s" #X0 ", 0xc\n\t"                                                                      \
  PROC1(X0, 7, b0, X1) PROC1(X2, 6, b1, X3) PROC1(X4, 5, b2, X5) PROC1(X6, 4, b3, X7)                      \
  PROC1(X8, 3, b4, X9) PROC1(X10, 2, b5, X11) PROC1(X12, 1, b6, X13) PROC1(X14, 0, b7, X15)                \
  "s_set_gpr_idx_off\n\t"

__global__ __launch_bounds__(NW * 64) void probe(const uint32_t* __restrict__ ent, int nbatch, int nstrip,
                                                 const float* __restrict__ Bt, float* __restrict__ out, int touch, int shared_stream) {
  extern __shared__ __align__(16) unsigned char lds[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  // fill both buffers once: slot s of buffer q holds Bt[(q * (W + 2) + s) * 128 + f]
  for (int t = threadIdx.x; t < (W + 2) * 2 * 64; t += NW * 64) {
    const int q = t / ((W + 2) * 64), r = t % ((W + 2) * 64), s = r / 64, l = r % 64;
    f32x2 v = {0.f, 0.f};
    if (s < W) v = *reinterpret_cast<const f32x2*>(Bt + ((size_t)(q * W + s) * 128 + l * 2));
    *reinterpret_cast<f32x2*>(lds + (s >> 1) * 2048 + q * 1024 + (s & 1) * 512 + l * 8) = v;
  }
  f32x32 accA, accB;
  for (int j = 0; j < 32; ++j) { accA[j] = 0.f; accB[j] = 0.f; }
  const unsigned char* base = reinterpret_cast<const unsigned char*>(ent) + ((size_t)(shared_stream ? 0 : blockIdx.x) * NW + wave) * (size_t)nstrip * nbatch * 64;
  uint32_t off = 0;
  const uint32_t mask = 0xfffffe00u;
  for (int s = 0; s < nstrip; ++s) {
    __syncthreads();
    uint32_t n = nbatch;
    if (touch && lane < 16) {                     // warm L2 with the lines a few strips ahead (nobody waits for the value)
      uint32_t t;
      asm volatile("global_load_dword %0, %1, off" : "=v"(t) : "v"(base + off + touch + lane * 64) : "memory");
    }
    const uint32_t lb = lane * 8 + (s & 1) * 1024;
    uint32_t a0, a1, a2, a3, a4, a5, a6, a7;
    f32x2 b0, b1, b2, b3, b4, b5, b6, b7;
    asm volatile(
        "s_load_dwordx16 s[36:51], %[base], %[off]\n\t"
        "s_add_u32 %[off], %[off], 64\n\t"
        "s_waitcnt lgkmcnt(0)\n\t"
        "1:\n\t"
        "s_load_dwordx16 s[52:67], %[base], %[off]\n\t"
        "s_add_u32 %[off], %[off], 64\n\t"
        PROCESS(36, 37, 38, 39, 40, 41, 42, 43, 44, 45, 46, 47, 48, 49, 50, 51)
        "s_sub_u32 %[n], %[n], 1\n\t"
        "s_cmp_eq_u32 %[n], 0\n\t"
        "s_cbranch_scc1 2f\n\t"
        "s_load_dwordx16 s[36:51], %[base], %[off]\n\t"
        "s_add_u32 %[off], %[off], 64\n\t"
        PROCESS(52, 53, 54, 55, 56, 57, 58, 59, 60, 61, 62, 63, 64, 65, 66, 67)
        "s_sub_u32 %[n], %[n], 1\n\t"
        "s_cmp_lg_u32 %[n], 0\n\t"
        "s_cbranch_scc1 1b\n\t"
        "2:\n\t"
        "s_waitcnt lgkmcnt(0)\n\t"
        "s_sub_u32 %[off], %[off], 64\n\t"
        : "+{v[64:95]}"(accA), "+{v[96:127]}"(accB), [off] "+s"(off), [n] "+s"(n),
          [a0] "=&v"(a0), [a1] "=&v"(a1), [a2] "=&v"(a2), [a3] "=&v"(a3), [a4] "=&v"(a4), [a5] "=&v"(a5), [a6] "=&v"(a6), [a7] "=&v"(a7),
          [b0] "=&v"(b0), [b1] "=&v"(b1), [b2] "=&v"(b2), [b3] "=&v"(b3), [b4] "=&v"(b4), [b5] "=&v"(b5), [b6] "=&v"(b6), [b7] "=&v"(b7)
        : [base] "s"(base), [mask] "v"(mask), [lb] "v"(lb)
        : "s36", "s37", "s38", "s39", "s40", "s41", "s42", "s43", "s44", "s45", "s46", "s47", "s48", "s49", "s50", "s51",
          "s52", "s53", "s54", "s55", "s56", "s57", "s58", "s59", "s60", "s61", "s62", "s63", "s64", "s65", "s66", "s67",
          "scc", "memory");
  }
  if (blockIdx.x == 0) {
    for (int j = 0; j < 32; ++j) out[(wave * 64 + j) * 64 + lane] = accA[j];
    for (int j = 0; j < 32; ++j) out[(wave * 64 + 32 + j) * 64 + lane] = accB[j];
  }
}

int main(int argc, char** argv) {
  const int nwg = argc > 1 ? atoi(argv[1]) : 512, nstrip = argc > 2 ? atoi(argv[2]) : 64, nbatch = argc > 3 ? atoi(argv[3]) : 7;
  const int touch = argc > 4 ? atoi(argv[4]) : 0, shared_stream = argc > 5 ? atoi(argv[5]) : 0;
  const size_t nent = (size_t)nwg * NW * nstrip * nbatch * 8;
  std::vector<uint32_t> ent(nent * 2 + 64 + 4096);
  srand(1);
  for (size_t i = 0; i < nent; ++i) {
    const int slot = rand() % W, row = rand() % 32;
    float v = (float)(rand() % 1000) / 1000.f;
    ent[2 * i] = ((uint32_t)(slot >> 1) << 11) | ((uint32_t)(slot & 1) << 9) | (uint32_t)(row * 2);
    memcpy(&ent[2 * i + 1], &v, 4);
  }
  std::vector<float> Bt((size_t)2 * W * 128);
  for (auto& x : Bt) x = (float)(rand() % 2000) / 1000.f - 1.f;
  uint32_t* d_ent; float *d_B, *d_out;
  hipMalloc(&d_ent, ent.size() * 4); hipMalloc(&d_B, Bt.size() * 4); hipMalloc(&d_out, NW * 64 * 64 * 4);
  hipMemcpy(d_ent, ent.data(), ent.size() * 4, hipMemcpyHostToDevice);
  hipMemcpy(d_B, Bt.data(), Bt.size() * 4, hipMemcpyHostToDevice);
  hipFuncSetAttribute((const void*)probe, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  float best = 1e9f;
  for (int it = 0; it < 5; ++it) {
    hipEventRecord(e0);
    hipLaunchKernelGGL(probe, dim3(nwg), dim3(NW * 64), LDS_BYTES, 0, d_ent, nbatch, nstrip, d_B, d_out, touch, shared_stream);
    hipEventRecord(e1);
    hipError_t e = hipDeviceSynchronize();
    if (e != hipSuccess) { printf("launch failed: %s\n", hipGetErrorString(e)); return 1; }
    float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
  }
  std::vector<float> out(NW * 64 * 64);
  hipMemcpy(out.data(), d_out, out.size() * 4, hipMemcpyDeviceToHost);
  int bad = 0;
  for (int w = 0; w < NW; ++w) {
    std::vector<float> want(64 * 128, 0.f);       // [acc index 0..63][lane*2 + e] -> acc index = row*2 + e
    const uint32_t* e = ent.data() + ((size_t)(0 * NW + w) * nstrip * nbatch * 8) * 2;
    for (int s = 0; s < nstrip; ++s)
      for (int k = 0; k < nbatch * 8; ++k, e += 2) {
        const int slot = (int)(((e[0] >> 11) << 1) | ((e[0] >> 9) & 1)), row = (int)((e[0] & 0xff) / 2);
        float v; memcpy(&v, &e[1], 4);
        for (int l = 0; l < 64; ++l)
          for (int c = 0; c < 2; ++c) {
            float& a = want[(row * 2 + c) * 64 + l];
            a = fmaf(v, Bt[((size_t)((s & 1) * W + slot) * 128 + l * 2 + c)], a);
          }
      }
    for (int j = 0; j < 64; ++j) for (int l = 0; l < 64; ++l)
      if (out[(w * 64 + j) * 64 + l] != want[j * 64 + l]) { if (bad < 5) printf("wave %d acc %d lane %d: %g want %g\n", w, j, l, out[(w * 64 + j) * 64 + l], want[j * 64 + l]); ++bad; }
  }
  const double entries = (double)nent;
  printf("touch %d shared %d: %d WGs x %d waves, %d strips x %d batches: %.3f ms, %.1f G entries/s (x128 floats = %.1f T FMA/s); %.2f clk/CU per entry at 2.4 GHz over 256 CUs; check %s (%d bad)\n",
         touch, shared_stream, nwg, NW, nstrip, nbatch, best, entries / best / 1e6, entries * 128 / best / 1e9, best * 1e-3 * 2.4e9 * 256 / entries, bad ? "FAIL" : "ok", bad);
  return bad != 0;
}
