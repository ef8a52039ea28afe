// stamps of the 64-rows-per-wave forward kernel (attn_fwd_w64.hip built with -DLCV_ATTN_STAMPS; run with LCV_ATTN_FWD_W64=1): where an iteration's cycles go
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdarg>
#include <cstdint>
#include <cmath>
#include <vector>
extern "C" int lcv_attn_fwd(const void* q, const void* k, const void* v, void* o, float* lse, int64_t B, int64_t H, int64_t Nq,
                            int64_t Nk, int64_t q_sb, int64_t q_sn, int64_t q_sh, int64_t k_sb, int64_t k_sn, int64_t k_sh,
                            int64_t v_sb, int64_t v_sn, int64_t v_sh, int64_t o_sb, int64_t o_sn, int64_t o_sh, float scale,
                            void* stream);
extern "C" void attn_w64_set_stamps(unsigned long long* buf, int block);
void lcv_set_error(const char* fmt, ...) { va_list ap; va_start(ap, fmt); vfprintf(stderr, fmt, ap); va_end(ap); fputc('\n', stderr); }
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
int main(int argc, char** argv) {
  const int64_t B = 1, H = 32, N = argc > 1 ? atol(argv[1]) : 46800, D = 128;
  const size_t n = (size_t)B * N * H * D;
  unsigned short *q, *k, *v, *o;
  hipMalloc(&q, n * 2); hipMalloc(&k, n * 2); hipMalloc(&v, n * 2); hipMalloc(&o, n * 2);
  fill<<<(unsigned)((n + 255) / 256), 256>>>(q, n, 1u, 0.1275f);   // unit-scale scores (|q||k| d^-1/2 log2 e)
  fill<<<(unsigned)((n + 255) / 256), 256>>>(k, n, 2u, 1.0f);
  fill<<<(unsigned)((n + 255) / 256), 256>>>(v, n, 3u, 1.0f);
  hipDeviceSynchronize();
  const int64_t sn = H * D, sb = N * sn, sh = D;
  auto run = [&]() { return lcv_attn_fwd(q, k, v, o, nullptr, B, H, N, N, sb, sn, sh, sb, sn, sh, sb, sn, sh, sb, sn, sh, 0.6931471806f, nullptr); };
  unsigned long long* dbg; hipMalloc(&dbg, 512 * 8); hipMemset(dbg, 0, 512 * 8);
  attn_w64_set_stamps(nullptr, 0);
  for (int i = 0; i < 2; ++i) if (run()) return 1;
  hipDeviceSynchronize();
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipEventRecord(e0);
  for (int i = 0; i < 5; ++i) run();
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 5;
  printf("w64 kernel N=%ld: %.3f ms  %.1f TF/s (stamp build, stamps off)\n", (long)N, ms, 4.0 * N * N * H * D * B / ms / 1e9);
  attn_w64_set_stamps(dbg, 3000);
  run(); hipDeviceSynchronize();
  std::vector<unsigned long long> h(512);
  hipMemcpy(h.data(), dbg, 512 * 8, hipMemcpyDeviceToHost);
  for (int w = 0; w < 2; ++w) {
    printf("wave %d: per iteration [phase1 | vmcnt+barrier | phase2 | settle] total, then gap to the next iteration's first stamp\n", w ? 2 : 0);
    for (int it = 0; it < 7; ++it) {
      const unsigned long long* s = &h[w * 256 + it * 8];
      if (!s[0]) continue;
      printf("  it %d:", 200 + it);
      for (int i = 0; i < 4; ++i) printf(" %5llu", s[i + 1] - s[i]);
      printf(" | %5llu   +%llu\n", s[4] - s[0], h[w * 256 + (it + 1) * 8] - s[4]);
    }
  }
  printf("wave2 - wave0 at iteration 200 start: %lld cycles\n", (long long)(h[256] - h[0]));
  return 0;
}
