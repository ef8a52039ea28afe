// What does it cost ONE wave per SIMD to issue a global->LDS or global->VGPR piece between MFMAs?
// Loop body: 64 x v_mfma_f32_16x16x32_bf16 (independent accumulators), one filler behind every 8th MFMA (8 per iteration):
//   0 none | 1 global_load_lds_dwordx4 (saddr + 32-bit voffset; M0 set one MFMA earlier) | 2 global_load_dwordx4 -> VGPR (same addressing)
//   3 buffer_load_dwordx4 -> VGPR (offen) | 4 buffer_load_dwordx4 ... lds (offen) | 5 ds_write_b128 | 6 = 2 + 5 (8 loads + 8 LDS writes)
//   7 global_load_lds_dwordx4 with a 64-bit per-lane address (no saddr) | 8 ds_read_b128 (16 per iteration, for scale)
// Every load reads a 64 KiB L2-resident region (issue cost, not memory latency: vmcnt(0) once per iteration, after the MFMAs).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;

template <int B, int E, class F> __device__ __forceinline__ void sfor(F&& f) {
  if constexpr (B < E) { f(std::integral_constant<int, B>{}); sfor<B + 1, E>(f); }
}

template <int MODE>
__global__ __launch_bounds__(256) void k(const char* src, float* out, int iters) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  f32x4 acc[16];
  for (int i = 0; i < 16; ++i) acc[i] = f32x4{0, 0, 0, 0};
  bf16x8 a, b;
  for (int j = 0; j < 8; ++j) { a[j] = (__bf16)(1.0f + lane * 1e-3f); b[j] = (__bf16)0.5f; }
  const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)smem + wave * 8192u;
  unsigned voff = (unsigned)((lane >> 3) * 512 + (lane & 7) * 16 + wave * 8192);   // 8 rows x 128 B, 512-B pitch
  const unsigned long long sv = (unsigned long long)src;
  const char* sbase = (const char*)(((unsigned long long)__builtin_amdgcn_readfirstlane((unsigned)(sv >> 32)) << 32) |
                                    __builtin_amdgcn_readfirstlane((unsigned)sv));
  u32x4 rsrc;   // raw buffer descriptor over the region
  rsrc[0] = (unsigned)sv; rsrc[1] = (unsigned)(sv >> 32) & 0xffffu; rsrc[2] = 1u << 20; rsrc[3] = 0x00020000u;
  rsrc[0] = __builtin_amdgcn_readfirstlane(rsrc[0]); rsrc[1] = __builtin_amdgcn_readfirstlane(rsrc[1]);
  rsrc[2] = __builtin_amdgcn_readfirstlane(rsrc[2]); rsrc[3] = __builtin_amdgcn_readfirstlane(rsrc[3]);
  const char* vaddr = src + voff;
  f32x4 st[8];
  for (int i = 0; i < 8; ++i) st[i] = f32x4{1.f * lane, 2, 3, 4};
  unsigned wr_addr = lds0 + lane * 16;
  for (int it = 0; it < iters; ++it) {
    sfor<0, 64>([&](auto g_c) {
      constexpr int g = decltype(g_c)::value;
      constexpr int t = g >> 3;
      constexpr bool fill = (g & 7) == 4;
      if constexpr (fill && (MODE == 1 || MODE == 4 || MODE == 7)) asm volatile("s_mov_b32 m0, %0" ::"s"(lds0 + t * 1024u));
      asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc[g & 15]) : "v"(a), "v"(b));
      if constexpr (fill) {
        if constexpr (MODE == 1) asm volatile("global_load_lds_dwordx4 %0, %1" ::"v"(voff), "s"(sbase) : "memory");
        if constexpr (MODE == 2 || MODE == 6) asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(st[t]) : "v"(voff), "s"(sbase) : "memory");
        if constexpr (MODE == 3) asm volatile("buffer_load_dwordx4 %0, %1, %2, 0 offen" : "=v"(st[t]) : "v"(voff), "s"(rsrc) : "memory");
        if constexpr (MODE == 4) asm volatile("buffer_load_dwordx4 %0, %1, 0 offen lds" ::"v"(voff), "s"(rsrc) : "memory");
        if constexpr (MODE == 7) asm volatile("global_load_lds_dwordx4 %0, off" ::"v"(vaddr) : "memory");
      }
      if constexpr ((g & 7) == 0 && (MODE == 5 || MODE == 6)) asm volatile("ds_write_b128 %0, %1" ::"v"(wr_addr), "v"(st[t]) : "memory");
      if constexpr ((g & 3) == 1 && MODE == 8) asm volatile("ds_read_b128 %0, %1" : "=v"(st[t]) : "v"(wr_addr) : "memory");
    });
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    // loads into st[] have landed; keep them live
    if constexpr (MODE == 2 || MODE == 3 || MODE == 6 || MODE == 8)
    {
      asm volatile("" : "+v"(st[0]), "+v"(st[1]), "+v"(st[2]), "+v"(st[3]));
      asm volatile("" : "+v"(st[4]), "+v"(st[5]), "+v"(st[6]), "+v"(st[7]));
    }
  }
  float r = 0;
  for (int i = 0; i < 16; ++i) r += acc[i][0];
  for (int i = 0; i < 8; ++i) r += st[i][0];
  if (r == 123.456f) out[threadIdx.x] = r;
}

template <typename F> float timeit(F f) {
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  f(500); (void)hipDeviceSynchronize();
  (void)hipEventRecord(e0); f(20000); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1); return ms * 1e6f / 20000;
}
int main() {
  float* out; (void)hipMalloc(&out, 4096);
  char* src; (void)hipMalloc(&src, 1 << 20); (void)hipMemset(src, 0, 1 << 20);
  printf("64 MFMA 16x16x32 per iteration, 1 wave/SIMD, 256 workgroups; filler behind every 8th MFMA (8 per iteration)\n");
  float base = 0;
#define RUN(M, name) { auto kk = k<M>; (void)hipFuncSetAttribute((const void*)kk, hipFuncAttributeMaxDynamicSharedMemorySize, 65536); \
    float ns = timeit([&](int it) { hipLaunchKernelGGL(kk, dim3(256), dim3(256), 65536, 0, src, out, it); }); if (M == 0) base = ns; \
    printf("  %-64s %7.1f ns/iter  (+%5.1f ns per filler = %4.0f MFMA-cycles at the bare loop's 16 cyc/MFMA)\n", name, ns, (ns - base) / (M == 8 ? 16 : 8), (ns - base) / (M == 8 ? 16 : 8) / (base / 1024)); }
  for (int rep = 0; rep < 2; ++rep) {
    RUN(0, "0 none")
    RUN(1, "1 global_load_lds_dwordx4 saddr + voffset")
    RUN(7, "7 global_load_lds_dwordx4 64-bit vaddr")
    RUN(4, "4 buffer_load_dwordx4 offen lds")
    RUN(2, "2 global_load_dwordx4 -> VGPR")
    RUN(3, "3 buffer_load_dwordx4 offen -> VGPR")
    RUN(5, "5 ds_write_b128")
    RUN(6, "6 global_load_dwordx4 -> VGPR + ds_write_b128 (8 + 8)")
    RUN(8, "8 ds_read_b128 (16 per iteration)")
  }
  if (hipDeviceSynchronize() != hipSuccess) { printf("FAILED\n"); return 1; }
  return 0;
}
