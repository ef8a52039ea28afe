// GEMM lab: per-phase barrier wait / work stamps of the 8-phase kernel on a K3-shaped projection
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdarg>
#include <cstdint>
#include <vector>
extern "C" int lcv_gemm_nt(const void* a, const void* w, const void* bias, const void* a2, const void* w2, void* c, int64_t M,
                           int64_t N, int64_t K, int64_t K2, int64_t lda, int64_t ldw, int64_t lda2, int64_t ldw2, int64_t ldc,
                           int epilogue, int out_f32, const void* resid, const float* mod, int64_t rows_per_frame,
                           int64_t mod_stride, int64_t gate_off, void* stream);
extern __device__ unsigned long long* g_gdbg;
void lcv_set_error(const char* fmt, ...) { va_list ap; va_start(ap, fmt); vfprintf(stderr, fmt, ap); va_end(ap); fputc('\n', stderr); }
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
  const int64_t M = 46800, N = argc > 1 ? atol(argv[1]) : 4096, K = argc > 2 ? atol(argv[2]) : 4096;
  unsigned short *a, *w, *b, *c;
  (void)hipMalloc(&a, M * K * 2); (void)hipMalloc(&w, N * K * 2); (void)hipMalloc(&b, N * 2); (void)hipMalloc(&c, M * N * 2);
  fill<<<(unsigned)((M * K + 255) / 256), 256>>>(a, M * K, 1u, 1.0f);
  fill<<<(unsigned)((N * K + 255) / 256), 256>>>(w, N * K, 2u, 0.02f);
  fill<<<(unsigned)((N + 255) / 256), 256>>>(b, N, 3u, 1.0f);
  (void)hipDeviceSynchronize();
  auto run = [&]() { return lcv_gemm_nt(a, w, b, nullptr, nullptr, c, M, N, K, 0, K, K, 0, 0, N, 0, 0, nullptr, nullptr, 1, 0, 0, nullptr); };
  unsigned long long* dbg; (void)hipMalloc(&dbg, 2 * 512 * 8);
  const char* modes[] = {"8", "9"};
  for (const char* m : modes) {
    setenv("LCV_GEMM_TILE", m, 1);
    unsigned long long* null = nullptr;
    (void)hipMemcpyToSymbol(HIP_SYMBOL(g_gdbg), &null, sizeof(null));
    for (int i = 0; i < 3; ++i) if (run()) return 1;
    (void)hipDeviceSynchronize();
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    (void)hipEventRecord(e0);
    for (int i = 0; i < 10; ++i) run();
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1); ms /= 10;
    printf("mode %s M=%ld N=%ld K=%ld: %.3f ms  %.1f TF/s\n", m, (long)M, (long)N, (long)K, ms, 2.0 * M * N * K / ms / 1e9);
    (void)hipMemset(dbg, 0, 2 * 512 * 8);
    (void)hipMemcpyToSymbol(HIP_SYMBOL(g_gdbg), &dbg, sizeof(dbg));
    run(); (void)hipDeviceSynchronize();
    std::vector<unsigned long long> h(1024);
    (void)hipMemcpy(h.data(), dbg, 1024 * 8, hipMemcpyDeviceToHost);
    for (int wv = 0; wv < 2; ++wv) {
      printf("  wave %d [bar1-wait lgkm mfma-issue bar2-wait loads]x phases:", wv ? 4 : 0);
      for (int i = 200; i < 240 && h[wv * 512 + i + 1]; ++i) printf(" %llu", h[wv * 512 + i + 1] - h[wv * 512 + i]);
      printf("\n");
    }
    // whole-tile anatomy: first stamp to last stamp of tile 0 of that block
    int n0 = 0; while (n0 < 511 && h[n0 + 1]) ++n0;
    printf("  wave 0: %d stamps, first->last %llu cycles\n", n0, h[n0] - h[0]);
  }
  return 0;
}
