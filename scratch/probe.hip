#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
// LDS: 4 rows x 16 cols block, row stride 64 bytes (32 elems); value = row*100+col
__global__ void probe_tr(short* out) {
  __shared__ __attribute__((aligned(16))) short lds[16*32];
  for (int i = threadIdx.x; i < 16*32; i += 64) lds[i] = (short)((i/32)*100 + (i%32));
  __syncthreads();
  int lane = threadIdx.x; int g = lane >> 4; int i = lane & 15; int q = i >> 2, p = i & 3;
  // group g reads block rows 4g..4g+3 (so groups differ), cols 0..15
  const short* addr = lds + (4*g + q)*32 + 4*p;
  s16x4 t = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)addr);
  for (int e = 0; e < 4; ++e) out[lane*4+e] = t[e];
}
// MFMA check: A[32x16], B[16x32] integer valued; D = A*B; dump acc regs
__global__ void probe_mfma(float* out) {
  int lane = threadIdx.x; int r = lane & 31, h = lane >> 5;
  bf16x8 a, b;
  for (int j = 0; j < 8; ++j) {
    int k = 8*h + j;
    a[j] = (__bf16)(float)((r == 3 && k == 5) ? 1 : 0);   // A[3][5] = 1
    b[j] = (__bf16)(float)((k == 5) ? (r + 1) : 0);       // B[5][c] = c+1
  }
  f32x16 acc = {0};
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc, 0, 0, 0);
  for (int e = 0; e < 16; ++e) out[lane*16+e] = acc[e];
}
int main() {
  short* d; hipMalloc(&d, 64*4*2); short h[256];
  hipLaunchKernelGGL(probe_tr, dim3(1), dim3(64), 0, 0, d); hipMemcpy(h, d, 512, hipMemcpyDeviceToHost);
  for (int l = 0; l < 64; ++l) { printf("lane %2d: %4d %4d %4d %4d\n", l, h[l*4], h[l*4+1], h[l*4+2], h[l*4+3]); }
  float* f; hipMalloc(&f, 64*16*4); float hf[1024];
  hipLaunchKernelGGL(probe_mfma, dim3(1), dim3(64), 0, 0, f); hipMemcpy(hf, f, 4096, hipMemcpyDeviceToHost);
  // expected D[3][c] = c+1 ; find where nonzero
  for (int l = 0; l < 64; ++l) for (int e = 0; e < 16; ++e) if (hf[l*16+e] != 0) printf("lane %d reg %d = %g\n", l, e, hf[l*16+e]);
  return 0;
}
