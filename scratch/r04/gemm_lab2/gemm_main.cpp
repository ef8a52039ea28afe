// lcv_gemm_nt (bias epilogue) at the K3 projection shapes, M = 93 600: ms, TF/s and a checksum of C (bitwise comparison of builds)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <cmath>
#include <cstring>
extern "C" int lcv_gemm_nt(const void* a, const void* w, const void* bias, const void* a2, const void* w2, void* c, int64_t M, int64_t N,
                           int64_t K, int64_t K2, int64_t lda, int64_t ldw, int64_t lda2, int64_t ldw2, int64_t ldc, int epilogue,
                           int out_f32, const void* resid, const float* mod, int64_t rows_per_frame, int64_t mod_stride, int64_t gate_off,
                           void* stream);
extern "C" const char* lcv_last_error(void);
__global__ void fill(unsigned short* p, size_t n, unsigned seed, float mul) {
  size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  if (i >= n) return;
  unsigned x = (unsigned)i * 2654435761u + seed; x ^= x >> 16; x *= 0x85ebca6bu; x ^= x >> 13; x *= 0xc2b2ae35u; x ^= x >> 16;
  unsigned y = x * 1664525u + 1013904223u;
  float u1 = ((x >> 8) + 1) * (1.0f / 16777217.0f), u2 = (y >> 8) * (1.0f / 16777216.0f);
  float g = sqrtf(-2.0f * logf(u1)) * cosf(6.2831853f * u2) * mul;
  unsigned bits = __float_as_uint(g);
  p[i] = (unsigned short)((bits + 0x7fff + ((bits >> 16) & 1)) >> 16);
}
__global__ void checksum(const unsigned short* p, size_t n, unsigned long long* out) {
  size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  if (i < n) atomicAdd(out, (unsigned long long)p[i] * (unsigned long long)((i % 65521) + 1));
}
int main(int argc, char** argv) {
  const char* label = argc > 1 ? argv[1] : "gemm";
  const int64_t M = argc > 2 ? atol(argv[2]) : 93600;
  struct { const char* name; int64_t N, K; } shapes[] = {{"qkv", 12288, 4096}, {"proj", 4096, 4096}, {"w13", 22016, 4096}, {"w2", 4096, 11008}};
  printf("%-22s", label);
  for (auto& sh : shapes) {
    const int64_t N = sh.N, K = sh.K;
    unsigned short *a, *w, *b, *c;
    hipMalloc(&a, (size_t)M * K * 2); hipMalloc(&w, (size_t)N * K * 2); hipMalloc(&b, (size_t)N * 2); hipMalloc(&c, (size_t)M * N * 2);
    fill<<<(unsigned)(((size_t)M * K + 255) / 256), 256>>>(a, (size_t)M * K, 11u, 1.0f);
    fill<<<(unsigned)(((size_t)N * K + 255) / 256), 256>>>(w, (size_t)N * K, 12u, 0.02f);
    fill<<<(unsigned)((N + 255) / 256), 256>>>(b, (size_t)N, 13u, 1.0f);
    auto run = [&]() { return lcv_gemm_nt(a, w, b, nullptr, nullptr, c, M, N, K, 0, K, K, 0, 0, N, 0, 0, nullptr, nullptr, 0, 0, 0, nullptr); };
    for (int i = 0; i < 3; ++i) if (run()) { fprintf(stderr, "gemm failed: %s\n", lcv_last_error()); return 1; }
    hipDeviceSynchronize();
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float best = 1e9f;
    for (int r = 0; r < 3; ++r) {
      hipEventRecord(e0);
      for (int i = 0; i < 6; ++i) run();
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 6; best = ms < best ? ms : best;
    }
    unsigned long long* cs; hipMalloc(&cs, 8); hipMemset(cs, 0, 8);
    checksum<<<(unsigned)(((size_t)M * N + 255) / 256), 256>>>(c, (size_t)M * N, cs);
    unsigned long long hcs; hipMemcpy(&hcs, cs, 8, hipMemcpyDeviceToHost);
    printf(" | %s %.3f ms %4.0f TF/s %08llx", sh.name, best, 2.0 * M * N * K / best / 1e9, hcs & 0xffffffffull);
    hipFree(a); hipFree(w); hipFree(b); hipFree(c); hipFree(cs);
  }
  printf("\n");
  return 0;
}
