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
