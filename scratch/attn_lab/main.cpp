// attention lab: times the variants of attn_pp.hip on K3-shaped random data and dumps per-stage s_memtime stamps
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdarg>
#include <cstdint>
#include <vector>
#include <cstring>
extern "C" int lcv_attn_fwd(const void* q, const void* k, const void* v, void* o, float* lse, int64_t B, int64_t H, int64_t Nq,
                            int64_t Nk, int64_t q_sb, int64_t q_sn, int64_t q_sh, int64_t k_sb, int64_t k_sn, int64_t k_sh,
                            int64_t v_sb, int64_t v_sn, int64_t v_sh, int64_t o_sb, int64_t o_sn, int64_t o_sh, float scale,
                            void* stream);
extern __device__ unsigned long long* g_dbg;
extern __device__ int g_dbg_block;
void lcv_set_error(const char* fmt, ...) { va_list ap; va_start(ap, fmt); vfprintf(stderr, fmt, ap); va_end(ap); fputc('\n', stderr); }
__global__ void fill(unsigned short* p, size_t n, unsigned seed) {
  size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  if (i >= n) return;
  unsigned x = (unsigned)i * 2654435761u + seed; x ^= x >> 16; x *= 0x85ebca6bu; x ^= x >> 13; x *= 0xc2b2ae35u; x ^= x >> 16;
  unsigned y = x * 1664525u + 1013904223u;
  float u1 = ((x >> 8) + 1) * (1.0f / 16777217.0f), u2 = (y >> 8) * (1.0f / 16777216.0f);
  float g = sqrtf(-2.0f * logf(u1)) * cosf(6.2831853f * u2);
  unsigned bits = __float_as_uint(g);
  p[i] = (unsigned short)((bits + 0x7fff + ((bits >> 16) & 1)) >> 16);
}
int main(int argc, char** argv) {
  const int64_t B = 1, H = 32, N = argc > 1 ? atol(argv[1]) : 46800, D = 128;
  const size_t n = (size_t)B * N * 3 * H * D;
  unsigned short *qkv, *o, *o_ref;
  hipMalloc(&qkv, n * 2); hipMalloc(&o, (size_t)B * N * H * D * 2); hipMalloc(&o_ref, (size_t)B * N * H * D * 2);
  fill<<<(unsigned)((n + 255) / 256), 256>>>(qkv, n, 12345u);
  hipDeviceSynchronize();
  const int64_t sn = 3 * H * D, sb = N * sn, sh = D;
  auto run = [&](void* out) {
    return lcv_attn_fwd(qkv, qkv + H * D, qkv + 2 * H * D, out, nullptr, B, H, N, N, sb, sn, sh, sb, sn, sh, sb, sn, sh, N * H * D,
                        H * D, D, 0.08838834764f, nullptr);
  };
  unsigned long long* dbg; hipMalloc(&dbg, 512 * 8); hipMemset(dbg, 0, 512 * 8);
  const char* vars[] = {"1", "1", "1", "3", "3", "3", "3", "3"};
  const char* prios[] = {"00", "30", "31", "00", "30", "31", "21", "03"};
  int vi = 0;
  for (const char* v : vars) {
    setenv("LCV_ATTN_VAR", v, 1);
    setenv("LCV_PRIO_STAGE", prios[vi], 1);
    printf("prio_stage(sm,mm)=%s ", prios[vi]); ++vi;
    void* outp = (v[0] == '1') ? (void*)o_ref : (void*)o;
    unsigned long long* null = nullptr;
    hipMemcpyToSymbol(HIP_SYMBOL(g_dbg), &null, sizeof(null));
    for (int i = 0; i < 2; ++i) if (run(outp)) return 1;
    hipDeviceSynchronize();
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    const int reps = 5;
    for (int i = 0; i < reps; ++i) run(outp);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= reps;
    printf("VAR=%s N=%ld: %.3f ms  %.1f TF/s\n", v, (long)N, ms, 4.0 * N * N * H * D * B / ms / 1e9);
    // instrumented launch (block in the middle of the grid)
    int blk = 3000 < (int)(((N + 255) / 256) * H) ? 3000 : 0;
    hipMemcpyToSymbol(HIP_SYMBOL(g_dbg_block), &blk, sizeof(blk));
    hipMemcpyToSymbol(HIP_SYMBOL(g_dbg), &dbg, sizeof(dbg));
    hipMemset(dbg, 0, 512 * 8);
    run(outp); hipDeviceSynchronize();
    std::vector<unsigned long long> h(512);
    hipMemcpy(h.data(), dbg, 512 * 8, hipMemcpyDeviceToHost);
    for (int w = 0; w < 2; ++w) {
      printf("  wave %d stage deltas (cycles):", w ? 4 : 0);
      for (int i = 60; i < 96 && h[w * 256 + i + 1]; ++i) printf(" %llu", h[w * 256 + i + 1] - h[w * 256 + i]);
      printf("\n");
    }
    printf("  wave4 - wave0 offset at stamp 60: %lld\n", (long long)(h[256 + 60] - h[60]));
  }
  // compare outputs
  std::vector<unsigned short> a((size_t)N * H * D), b((size_t)N * H * D);
  hipMemcpy(a.data(), o_ref, a.size() * 2, hipMemcpyDeviceToHost); hipMemcpy(b.data(), o, b.size() * 2, hipMemcpyDeviceToHost);
  size_t diff = 0; for (size_t i = 0; i < a.size(); ++i) diff += a[i] != b[i];
  printf("outputs differing: %zu of %zu\n", diff, a.size());
  return 0;
}
