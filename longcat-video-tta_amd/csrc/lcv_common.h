// Shared device/host helpers for liblcv_hip.so (gfx950 only; wave = 64).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>
#include "../../include/lcv_hip.h"

typedef unsigned short bf16_t;  // raw storage
typedef __attribute__((ext_vector_type(2))) unsigned short u16x2;
typedef __attribute__((ext_vector_type(4))) unsigned short u16x4;
typedef __attribute__((ext_vector_type(8))) unsigned short u16x8;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;

#define LCV_WAVE 64

// ---- error plumbing (host) ----
void lcv_set_error(const char* fmt, ...);
#define LCV_CHECK_ARG(cond, ...)            \
  do {                                      \
    if (!(cond)) {                          \
      lcv_set_error(__VA_ARGS__);           \
      return LCV_EINVAL;                    \
    }                                       \
  } while (0)
#define LCV_LAUNCH_CHECK(name)                                             \
  do {                                                                     \
    hipError_t e__ = hipGetLastError();                                    \
    if (e__ != hipSuccess) {                                               \
      lcv_set_error("%s: launch failed: %s", name, hipGetErrorString(e__)); \
      return LCV_ELAUNCH;                                                  \
    }                                                                      \
  } while (0)

// ---- A/B knobs (host) ----
// The launchers' LCV_* environment variables are read ONCE, when the library is first used, into a table in lib.hip (the list is
// part of the documented interface: include/lcv_hip.h).  `lcv_knob(name)` returns the value or nullptr when it is unset;
// `lcv_knobs_reload()` (C ABI) reads the environment again - for tests and A/B harnesses that flip a knob inside one process.
const char* lcv_knob(const char* name);

// ---- bf16 <-> f32 (device) ----
__device__ __forceinline__ float bf2f(bf16_t u) {
  return __builtin_bit_cast(float, (unsigned int)u << 16);
}
// round-to-nearest-even via the hardware convert (keeps NaN a NaN)
__device__ __forceinline__ bf16_t f2bf(float f) {
  __bf16 b = (__bf16)f;
  return __builtin_bit_cast(bf16_t, b);
}
__device__ __forceinline__ float bfround(float f) { return bf2f(f2bf(f)); }

__device__ __forceinline__ void unpack8(const u16x8& v, float (&f)[8]) {
#pragma unroll
  for (int i = 0; i < 8; ++i) f[i] = bf2f(v[i]);
}
__device__ __forceinline__ u16x8 pack8(const float (&f)[8]) {
  u16x8 v;
#pragma unroll
  for (int i = 0; i < 8; ++i) v[i] = f2bf(f[i]);
  return v;
}

// ---- wave reductions ----
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

__device__ __forceinline__ float silu_f(float x) { return x / (1.0f + __expf(-x)); }

// ---- LDS-DMA from inline asm (device) ----
// 16 bytes per lane from a per-lane global address to LDS byte address `lds_dst` + 16 * lane (wave-uniform `lds_dst`).
// Why not __builtin_amdgcn_global_load_lds: hipcc books a builtin LDS-DMA as a pending LDS write and parks an
// `s_waitcnt vmcnt(0)` in front of the NEXT ds_read - in a double-buffered loop that waits for the tile just requested before
// the current one is even read (the attention backward passes ran {issue DMA; wait for it; compute} for a round and a half:
// SQ_WAIT_ANY 0.52 of the wave cycles).  An asm DMA is invisible to that bookkeeping: the CALLER waits (`lcv_dma_wait_all`)
// before the barrier that publishes the tile.  M0 carries the LDS destination and is restored: hipcc owns it.
__device__ __forceinline__ void lcv_lds_dma16(const void* gsrc, unsigned lds_dst) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep)
               : "v"(gsrc), "s"(lds_dst)
               : "memory");
}
// the same with a wave-uniform 64-bit base (SGPR pair) + a 32-bit per-lane byte offset: no 64-bit vector address arithmetic
__device__ __forceinline__ void lcv_lds_dma16_sv(unsigned voff, const char* sbase, unsigned lds_dst) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
               : "=&s"(keep)
               : "v"(voff), "s"(sbase), "s"(lds_dst)
               : "memory");
}
// 4 bytes per lane (256 B per wave-instruction), same addressing
__device__ __forceinline__ void lcv_lds_dma4_sv(unsigned voff, const char* sbase, unsigned lds_dst) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dword %1, %2\n\ts_mov_b32 m0, %0"
               : "=&s"(keep)
               : "v"(voff), "s"(sbase), "s"(lds_dst)
               : "memory");
}
// readfirstlane makes the uniformity of a pointer provable, so that an "s" asm operand gets SGPRs
__device__ __forceinline__ const char* lcv_uniform_ptr(const void* ptr) {
  const unsigned long long v = (unsigned long long)ptr;
  const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
  return (const char*)(((unsigned long long)hi << 32) | lo);
}
__device__ __forceinline__ void lcv_dma_wait_all() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
