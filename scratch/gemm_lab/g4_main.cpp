// Lab for csrc/gemm4w.h: one binary per -DG4_LAB_* combination, each timing the 4-wave kernel on a K3 projection shape
// (results are wrong on purpose when an ingredient is removed; this measures what it costs).
#include "/tmp/g4pre.h"
#include "gemm4w.h"
#include <cstdio>
#include <cstdlib>
void lcv_set_error(const char* fmt, ...) {}
__global__ void fill(unsigned short* p, size_t n, unsigned seed, float scale) {
  size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  if (i >= n) return;
  unsigned x = (unsigned)i * 2654435761u + seed; x ^= x >> 16; x *= 0x85ebca6bu; x ^= x >> 13; x *= 0xc2b2ae35u; x ^= x >> 16;
  unsigned y = x * 1664525u + 1013904223u;
  float u1 = ((x >> 8) + 1) * (1.0f / 16777217.0f), u2 = (y >> 8) * (1.0f / 16777216.0f);
  float g = sqrtf(-2.0f * logf(u1)) * cosf(6.2831853f * u2) * scale;
  unsigned bits = __float_as_uint(g);
  p[i] = (unsigned short)((bits + 0x7fff + ((bits >> 16) & 1)) >> 16);
}
int main(int argc, char** argv) {
  const int64_t M = 93600, N = argc > 1 ? atol(argv[1]) : 4096, K = argc > 2 ? atol(argv[2]) : 4096;
  unsigned short *a, *w, *b, *c;
  (void)hipMalloc(&a, M * K * 2); (void)hipMalloc(&w, N * K * 2); (void)hipMalloc(&b, N * 2); (void)hipMalloc(&c, M * N * 4 + (1 << 20));
  fill<<<(unsigned)((M * K + 255) / 256), 256>>>(a, M * K, 1u, 1.0f);
  fill<<<(unsigned)((N * K + 255) / 256), 256>>>(w, N * K, 2u, 0.02f);
  fill<<<(unsigned)((N + 255) / 256), 256>>>(b, N, 3u, 1.0f);
  (void)hipDeviceSynchronize();
  GemmParams p{};
  p.a = a; p.w = w; p.bias = b; p.c = c; p.M = M; p.N = N; p.nk1 = (int)(K / 64); p.nk2 = 0;
  p.lda = K; p.ldw = K; p.ldc = N; p.rows_per_frame = 1;
  p.tiles_m = (int)((M + 255) / 256); p.tiles_n = (int)((N + 255) / 256); p.group_m = 6;
  p.vid_begin = 0; p.vid_count = p.tiles_m * p.tiles_n; p.splitk = 1;
  #ifndef G4_LAB_NW
#define G4_LAB_NW 4
#endif
  auto kern = gemm4w_nt_kernel<0, G4_LAB_NW>;
  (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
  auto run = [&]() { hipLaunchKernelGGL(kern, dim3(256), dim3(G4_LAB_NW * 64), 131072, 0, p); };
  for (int i = 0; i < 3; ++i) run();
  if (hipDeviceSynchronize() != hipSuccess) { printf("launch failed\n"); return 1; }
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  (void)hipEventRecord(e0);
  for (int i = 0; i < 10; ++i) run();
  (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1); ms /= 10;
  printf("%-14s N=%ld K=%ld: %.3f ms  %.1f TF/s\n", G4_LAB_NAME, (long)N, (long)K, ms, 2.0 * M * N * K / ms / 1e9);
  return 0;
}
