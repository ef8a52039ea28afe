// stamps of the software-pipelined pass A of the attention backward (attn_bwd_dkv3.hip built with -DLCV_DKV3_STAMPS): where an
// iteration's cycles go (phase Y | phase X | wait for the LDS-DMA | barrier), for waves 0 and 2 of one workgroup
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdarg>
#include <cstdint>
#include <cmath>
#include <vector>
int attn_bwd_dkv3_launch(const void* q, const void* k, const void* v, const void* d_o, const float* consts,
                         void* dk, void* dv, int accumulate_kv, int64_t B, int64_t H, int64_t Nq, int64_t Nk, int64_t q_sb,
                         int64_t q_sn, int64_t q_sh, int64_t k_sb, int64_t k_sn, int64_t k_sh, int64_t v_sb, int64_t v_sn,
                         int64_t v_sh, int64_t o_sb, int64_t o_sn, int64_t o_sh, int64_t dk_sb, int64_t dk_sn, int64_t dk_sh,
                         int64_t dv_sb, int64_t dv_sn, int64_t dv_sh, float scale, hipStream_t s);
extern "C" void attn_dkv3_set_stamps(unsigned long long* buf, int block);
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
__global__ void fillf(float* p, size_t n, float v) { size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; if (i < n) p[i] = v; }
int main(int argc, char** argv) {
  const int64_t B = 1, H = 32, N = argc > 1 ? atol(argv[1]) : 14400, D = 128;
  const size_t n = (size_t)B * N * H * D;
  unsigned short *q, *k, *v, *d_o, *dk, *dv;
  hipMalloc(&q, n * 2); hipMalloc(&k, n * 2); hipMalloc(&v, n * 2); hipMalloc(&d_o, n * 2); hipMalloc(&dk, n * 2); hipMalloc(&dv, n * 2);
  fill<<<(unsigned)((n + 255) / 256), 256>>>(q, n, 1u, 0.1275f);
  fill<<<(unsigned)((n + 255) / 256), 256>>>(k, n, 2u, 1.0f);
  fill<<<(unsigned)((n + 255) / 256), 256>>>(v, n, 3u, 1.0f);
  fill<<<(unsigned)((n + 255) / 256), 256>>>(d_o, n, 4u, 1.0f);
  const int64_t Nqp = (N + 31) / 32 * 32;
  float* consts; hipMalloc(&consts, (size_t)B * H * 2 * Nqp * 4);
  // plausible row constants: -lse in log2 units around -16 (P ~ 2^-12 .. 2^-20), -delta small
  for (int64_t bh = 0; bh < B * H; ++bh) {
    fillf<<<(unsigned)((Nqp + 255) / 256), 256>>>(consts + bh * 2 * Nqp, Nqp, -16.0f);
    fillf<<<(unsigned)((Nqp + 255) / 256), 256>>>(consts + bh * 2 * Nqp + Nqp, Nqp, -0.01f);
  }
  hipDeviceSynchronize();
  const int64_t sn = H * D, sb = N * sn, sh = D;
  auto run = [&]() { return attn_bwd_dkv3_launch(q, k, v, d_o, consts, dk, dv, 0, B, H, N, N, sb, sn, sh, sb, sn, sh, sb, sn, sh, sb, sn, sh,
                                                sb, sn, sh, sb, sn, sh, 0.6931471806f, nullptr); };
  unsigned long long* dbg; hipMalloc(&dbg, 128 * 8); hipMemset(dbg, 0, 128 * 8);
  attn_dkv3_set_stamps(nullptr, 0);
  for (int i = 0; i < 2; ++i) if (run()) return 1;
  hipDeviceSynchronize();
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipEventRecord(e0);
  for (int i = 0; i < 5; ++i) run();
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 5;
  const double tiles = (double)((N + 31) / 32) * ((N + 127) / 128) * H;     // workgroup-tiles
  printf("dkv3 N=%ld: %.3f ms  %.1f TF/s executed (8 N^2 D H; stamp build, stamps off); %.0f ns per workgroup-tile on 256 CUs\n", (long)N, ms,
         8.0 * N * N * H * D * B / ms / 1e9, ms * 1e6 / (tiles / 256));
  attn_dkv3_set_stamps(dbg, 40);
  run(); hipDeviceSynchronize();
  std::vector<unsigned long long> h(128);
  hipMemcpy(h.data(), dbg, 128 * 8, hipMemcpyDeviceToHost);
  for (int w = 0; w < 2; ++w) {
    printf("wave %d: per iteration [phase Y | phase X | vmcnt | barrier] total, then gap to the next iteration's first stamp\n", w ? 2 : 0);
    for (int it = 0; it < 7; ++it) {
      const unsigned long long* s = &h[w * 64 + it * 8];
      if (!s[0]) continue;
      printf("  it %d:", 100 + it);
      for (int i = 0; i < 4; ++i) printf(" %5llu", s[i + 1] - s[i]);
      printf(" | %5llu   +%llu\n", s[4] - s[0], h[w * 64 + (it + 1) * 8] - s[4]);
    }
  }
  printf("wave2 - wave0 at iteration 100 start: %lld cycles\n", (long long)(h[64] - h[0]));
  return 0;
}
