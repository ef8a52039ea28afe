// Do an MFMA stream and a VALU stream of two waves on ONE SIMD overlap?  waves 0-3: MFMA loop, waves 4-7: VALU loop.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
__global__ __launch_bounds__(512) void k(float* out, int iters, int mode, int valu_kind) {
  const int wave = threadIdx.x >> 6;
  const bool do_mfma = (wave < 4) ? (mode & 1) : (mode & 4);
  const bool do_valu = (wave < 4) ? (mode & 8) : (mode & 2);
  float r = 0.f;
  if (do_mfma) {
    f32x16 acc[4];
    for (int i = 0; i < 4; ++i) for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
    bf16x8 a, b;
    for (int j = 0; j < 8; ++j) { a[j] = (__bf16)(1.0f + threadIdx.x * 1e-3f); b[j] = (__bf16)(0.5f); }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int u = 0; u < 16; ++u) acc[u & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[u & 3], 0, 0, 0);
    }
    for (int i = 0; i < 4; ++i) r += acc[i][0];
  }
  if (do_valu) {
    float x[16];
    for (int i = 0; i < 16; ++i) x[i] = threadIdx.x * 1e-4f + i;
    for (int it = 0; it < iters; ++it) {
      if (valu_kind == 0) {
#pragma unroll
        for (int u = 0; u < 64; ++u) x[u & 15] = __builtin_fmaf(x[u & 15], 0.999f, 0.001f);   // 64 v_fma
      } else {
#pragma unroll
        for (int u = 0; u < 32; ++u) { x[u & 15] = __builtin_amdgcn_exp2f(x[u & 15] * 0.5f - 1.0f); }  // 32 (mul/fma + exp)
      }
    }
    for (int i = 0; i < 16; ++i) r += x[i];
  }
  if (r == 123.456f) out[threadIdx.x] = r;
}
int main() {
  float* out; hipMalloc(&out, 4096);
  const int iters = 20000;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  struct { int mode; const char* name; } modes[] = {{1, "mfma(w0-3) only"}, {2, "valu(w4-7) only"}, {3, "mfma(w0-3) + valu(w4-7)"},
                                                     {5, "mfma(w0-3) + mfma(w4-7)"}, {10, "valu both"}, {12, "valu(w0-3) + mfma(w4-7)"}};
  for (int kind = 0; kind < 2; ++kind) {
    printf("VALU kind %d (%s)\n", kind, kind ? "32x{fma,exp}" : "64xfma");
    for (auto& m : modes) {
      k<<<256, 512>>>(out, 100, m.mode, kind); hipDeviceSynchronize();
      hipEventRecord(e0);
      k<<<256, 512>>>(out, iters, m.mode, kind);
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      printf("  %-28s %.3f ms  (%.1f ns/iter)\n", m.name, ms, ms * 1e6 / iters);
    }
  }
  return 0;
}
