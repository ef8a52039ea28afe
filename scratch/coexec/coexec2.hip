// (1) does s_setprio change the cross-wave MFMA/VALU serialisation?  (2) does ONE wave overlap its own MFMAs with interleaved VALU?
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
// mode: 0 = w0-3 MFMA, w4-7 VALU(exp mix);  prio_mfma / prio_valu = s_setprio value of each role
// mode: 1 = every wave: 16 MFMA with NV VALU ops interleaved after each MFMA (single stream), waves 4-7 idle if half==1
template <int NV, int KIND>
__global__ __launch_bounds__(512) void k(float* out, int iters, int mode, int prio_mfma, int prio_valu, int half) {
  const int wave = threadIdx.x >> 6;
  float r = 0.f;
  f32x16 acc[4];
  for (int i = 0; i < 4; ++i) for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
  bf16x8 a, b;
  for (int j = 0; j < 8; ++j) { a[j] = (__bf16)(1.0f + threadIdx.x * 1e-3f); b[j] = (__bf16)(0.5f); }
  float x[16];
  for (int i = 0; i < 16; ++i) x[i] = threadIdx.x * 1e-4f + i;
  if (mode == 0) {
    if (wave < 4) {
      if (prio_mfma == 1) __builtin_amdgcn_s_setprio(1); else if (prio_mfma == 2) __builtin_amdgcn_s_setprio(2); else if (prio_mfma == 3) __builtin_amdgcn_s_setprio(3);
      for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 16; ++u) acc[u & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[u & 3], 0, 0, 0);
      }
    } else {
      if (prio_valu == 1) __builtin_amdgcn_s_setprio(1); else if (prio_valu == 2) __builtin_amdgcn_s_setprio(2); else if (prio_valu == 3) __builtin_amdgcn_s_setprio(3);
      for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 32; ++u) x[u & 15] = __builtin_amdgcn_exp2f(x[u & 15] * 0.5f - 1.0f);
      }
    }
  } else {
    if (half == 0 || wave < 4) {
      for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 16; ++u) {
          acc[u & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[u & 3], 0, 0, 0);
#pragma unroll
          for (int v = 0; v < NV; ++v) {
            const int idx = (u * NV + v) & 15;
            if (KIND == 0) x[idx] = __builtin_fmaf(x[idx], 0.999f, 0.001f);
            else x[idx] = __builtin_amdgcn_exp2f(x[idx] * 0.5f - 1.0f);
          }
          __builtin_amdgcn_sched_barrier(0);
        }
      }
    }
  }
  for (int i = 0; i < 4; ++i) r += acc[i][0];
  for (int i = 0; i < 16; ++i) r += x[i];
  if (r == 123.456f) out[threadIdx.x] = r;
}
template <typename F> float timeit(F f) {
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  f(100); (void)hipDeviceSynchronize();
  (void)hipEventRecord(e0); f(20000); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1); return ms * 1e6f / 20000;
}
int main() {
  float* out; (void)hipMalloc(&out, 4096);
  printf("cross-wave: w0-3 16 MFMA/iter, w4-7 32 {fma,exp}/iter (alone: ~218 ns each)\n");
  for (int pm = 0; pm < 2; ++pm) for (int pv = 0; pv < 4; ++pv)
    printf("  prio mfma=%d valu=%d : %.1f ns/iter\n", pm, pv, timeit([&](int it) { k<0, 0><<<256, 512>>>(out, it, 0, pm, pv, 0); }));
  printf("single stream, 16 MFMA/iter (alone 250 ns) with NV VALU after each MFMA; 1 wave/SIMD (half=1) and 2 waves/SIMD\n");
#define RUN(NV, KIND) printf("  NV=%d kind=%s : 1 wave/SIMD %.1f ns   2 waves/SIMD %.1f ns/iter\n", NV, KIND ? "fma+exp" : "fma", \
      timeit([&](int it) { k<NV, KIND><<<256, 512>>>(out, it, 1, 0, 0, 1); }), timeit([&](int it) { k<NV, KIND><<<256, 512>>>(out, it, 1, 0, 0, 0); }));
  RUN(0, 0) RUN(2, 0) RUN(4, 0) RUN(6, 0) RUN(8, 0) RUN(1, 1) RUN(2, 1) RUN(3, 1) RUN(4, 1)
  return 0;
}
