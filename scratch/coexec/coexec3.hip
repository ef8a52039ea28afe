// What arrangement of an attention-like per-wave stream  [16 MFMA | softmax-like VALU block | 16 MFMA]  keeps the matrix pipe of
// a SIMD busiest with TWO waves per SIMD (512-thread workgroup, no LDS, no barriers: issue arbitration only)?
//   mode 0: both waves run the phases in lockstep
//   mode 1: waves 4-7 start half a period late (their VALU block falls under the partner's MFMA blocks)
//   mode 2: lockstep, s_setprio(P) around the VALU block
//   mode 3: staggered + s_setprio(P) around the VALU block
//   mode 4: software-pipelined single stream: the VALU of the block spread between the 32 MFMAs (NV per gap), lockstep
//   mode 5: as 4, staggered by half a period
// VALU block per tile: 32 v_exp_f32, 32 v_add_f32, 16 v_max3_f32, 16 v_cvt_pk (the forward kernel's steady-state tile).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

__device__ __forceinline__ void valu_block(float (&x)[32], float& acc, float& mx, unsigned (&pk)[16]) {
#pragma unroll
  for (int i = 0; i < 32; i += 2) mx = __builtin_fmaxf(__builtin_fmaxf(mx, x[i]), x[i + 1]);   // 16 v_max3
#pragma unroll
  for (int i = 0; i < 32; ++i) { x[i] = __builtin_amdgcn_exp2f(x[i]); acc += x[i]; }             // 32 exp + 32 add
#pragma unroll
  for (int i = 0; i < 16; ++i) asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(pk[i]) : "v"(x[2 * i]), "v"(x[2 * i + 1]));
}

template <int MODE, int PRIO>
__global__ __launch_bounds__(512) void k(float* out, int iters) {
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  f32x16 acc[4];
  for (int i = 0; i < 4; ++i) for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
  bf16x8 a, b;
  for (int j = 0; j < 8; ++j) { a[j] = (__bf16)(1.0f + threadIdx.x * 1e-3f); b[j] = (__bf16)(0.5f); }
  float x[32];
  for (int i = 0; i < 32; ++i) x[i] = -(threadIdx.x * 1e-4f + i * 0.01f);
  float sum = 0.f, mx = -1e30f;
  unsigned pk[16];
  for (int i = 0; i < 16; ++i) pk[i] = 0;
  const bool late = (MODE == 1 || MODE == 3 || MODE == 5) && wave >= 4;
  auto mfma16 = [&]() {
#pragma unroll
    for (int u = 0; u < 16; ++u) acc[u & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[u & 3], 0, 0, 0);
  };
  if (late) { mfma16(); __builtin_amdgcn_sched_barrier(0); }   // half a period of offset
  for (int it = 0; it < iters; ++it) {
    if (MODE < 4) {
      mfma16();
      __builtin_amdgcn_sched_barrier(0);
      if (MODE >= 2) __builtin_amdgcn_s_setprio(PRIO);
      valu_block(x, sum, mx, pk);
      if (MODE >= 2) __builtin_amdgcn_s_setprio(0);
      __builtin_amdgcn_sched_barrier(0);
      mfma16();
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int i = 0; i < 32; ++i) x[i] = x[i] * 0.5f - 1.0f;   // keep the values in range (stands for the next tile's scores)
    } else {
      // 32 gaps: gap g gets exp of x[g] + add, every other gap a max3, every other gap a cvt
#pragma unroll
      for (int g = 0; g < 32; ++g) {
        acc[g & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[g & 3], 0, 0, 0);
        x[g] = __builtin_amdgcn_exp2f(x[g] * 0.5f - 1.0f);
        sum += x[g];
        if (g & 1) {
          mx = __builtin_fmaxf(__builtin_fmaxf(mx, x[g - 1]), x[g]);
          asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(pk[g >> 1]) : "v"(x[g - 1]), "v"(x[g]));
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  }
  float r = sum + mx;
  for (int i = 0; i < 4; ++i) r += acc[i][0];
  for (int i = 0; i < 16; ++i) r += (float)pk[i];
  if (r == 123.456f) out[threadIdx.x] = r;
}

template <typename F> float timeit(F f) {
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  f(200); (void)hipDeviceSynchronize();
  (void)hipEventRecord(e0); f(20000); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1); return ms * 1e6f / 20000;
}
int main() {
  float* out; (void)hipMalloc(&out, 4096);
  printf("per iteration and wave: 32 MFMA 32x32x16 + {32 exp, 32 add (+32 fma in modes 0-3), 16 max3, 16 cvt_pk}; 2 waves / SIMD, 256 WGs\n");
  printf("matrix-pipe floor: 2 waves x 32 MFMA x 32 cycles = 2048 cycles per iteration\n");
#define RUN(M, P, name) printf("  %-58s %.1f ns/iter\n", name, timeit([&](int it) { k<M, P><<<256, 512>>>(out, it); }));
  for (int rep = 0; rep < 2; ++rep) {
    RUN(0, 0, "0 lockstep")
    RUN(1, 0, "1 staggered (waves 4-7 half a period late)")
    RUN(2, 1, "2 lockstep, setprio 1 around the VALU block")
    RUN(3, 1, "3 staggered, setprio 1 around the VALU block")
    RUN(3, 3, "3 staggered, setprio 3 around the VALU block")
    RUN(4, 0, "4 VALU spread between the MFMAs, lockstep")
    RUN(5, 0, "5 VALU spread between the MFMAs, staggered")
  }
  return 0;
}
