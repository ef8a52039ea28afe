// HBM-bound stages of the WAN-style causal 3-D VAE decoder on channels-last activations:
// the channel RMS norm (+SiLU) in front of every conv, and the row softmax of the per-frame
// single-head mid-block attention (head_dim 384: scores and PV go through the MFMA GEMM).
#include "lcv_common.h"

// y[c] = x[c] / max(||x||_2, 1e-12) * sqrt(C) * gamma[c]  (then SiLU); channels >= C (padding) are written as 0.
// HBM-bound: 4 B per element (bf16 in, bf16 out).  A pixel row of Cpad channels is Cpad / 8 sixteen-byte pieces; LPR lanes
// (the next power of two: 16 at the 128-channel top level, 32 at 256, 64 at 384 / 512) share one row, so a wave covers
// 64 / LPR consecutive rows and EVERY lane moves 16 bytes per instruction (the first form gave a whole wave to each row: at
// 128 channels 48 of 64 lanes idled and the 49x720p top level ran at 0.17 of the HBM peak).  Workgroups walk the rows with a
// grid stride.
template <int LPR>
__global__ __launch_bounds__(256) void vae_rmsnorm_kernel(const bf16_t* __restrict__ x,
                                                          const bf16_t* __restrict__ gamma,
                                                          bf16_t* __restrict__ y, int64_t rows, int C, int Cpad,
                                                          int apply_silu) {
  constexpr int RPW = 64 / LPR;                       // rows per wave
  const int lane = threadIdx.x & 63;
  const int sub = lane / LPR;                         // which of the wave's rows
  const int c = (lane % LPR) * 8;
  const bool live = c < Cpad;
  float g[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  if (live) unpack8(*reinterpret_cast<const u16x8*>(gamma + c), g);  // gamma is stored padded to Cpad
  const float sqrt_c = sqrtf((float)C);
  const int64_t wave_id = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int64_t n_waves = (int64_t)gridDim.x * 4;
  // LPR = 16 (<= 128 channels, the 720p level): two row groups per iteration, both loads issued before either is used (one group
  // in flight per wave left that level at 4.7 TB/s: 5.02 -> 4.69 ms); the wider rows lost with the second group (1.59 -> 1.84 ms)
  constexpr int U = LPR == 16 ? 2 : 1;
  const int64_t step = n_waves * RPW;
  for (int64_t r0 = wave_id * RPW; r0 < rows; r0 += U * step) {
    int64_t row[U];
    bool ok[U];
    float v[U][8];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      row[u] = r0 + u * step + sub;
      ok[u] = live && row[u] < rows;
#pragma unroll
      for (int i = 0; i < 8; ++i) v[u][i] = 0.f;
    }
    u16x8 raw[U];
#pragma unroll
    for (int u = 0; u < U; ++u)
      if (ok[u]) raw[u] = *reinterpret_cast<const u16x8*>(x + row[u] * Cpad + c);
#pragma unroll
    for (int u = 0; u < U; ++u) {
      if (ok[u]) unpack8(raw[u], v[u]);
      float ss = 0.f;
#pragma unroll
      for (int i = 0; i < 8; ++i)
        if (c + i < C) ss += v[u][i] * v[u][i];
#pragma unroll
      for (int o = LPR / 2; o > 0; o >>= 1) ss += __shfl_xor(ss, o, 64);   // stays inside the row's LPR lanes
      const float inv = sqrt_c / fmaxf(sqrtf(ss), 1e-12f);
      if (ok[u]) {
        float o8[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          float t = (c + i < C) ? v[u][i] * inv * g[i] : 0.f;
          if (apply_silu) t = silu_f(t);
          o8[i] = t;
        }
        *reinterpret_cast<u16x8*>(y + row[u] * Cpad + c) = pack8(o8);
      }
    }
  }
}

extern "C" int lcv_vae_rmsnorm_silu(const void* x, const void* gamma, void* y, int64_t rows, int64_t C,
                                    int64_t Cpad, int apply_silu, void* stream) {
  LCV_CHECK_ARG(x && gamma && y, "vae_rmsnorm_silu: null pointer");
  LCV_CHECK_ARG(Cpad % 8 == 0 && Cpad <= 512 && C <= Cpad && C > 0, "vae_rmsnorm_silu: C=%ld Cpad=%ld unsupported", (long)C, (long)Cpad);
  if (rows == 0) return LCV_OK;
  const int lpr = Cpad <= 128 ? 16 : (Cpad <= 256 ? 32 : 64);
  const int64_t rows_per_block = 4 * (64 / lpr);
  int64_t blocks = (rows + rows_per_block - 1) / rows_per_block;
  if (blocks > 256 * 32) blocks = 256 * 32;          // grid stride beyond 32 workgroups per CU
  const dim3 grid((unsigned)blocks), blk(256);
  hipStream_t st = (hipStream_t)stream;
  if (lpr == 16)
    hipLaunchKernelGGL(vae_rmsnorm_kernel<16>, grid, blk, 0, st, (const bf16_t*)x, (const bf16_t*)gamma, (bf16_t*)y, rows, (int)C, (int)Cpad, apply_silu);
  else if (lpr == 32)
    hipLaunchKernelGGL(vae_rmsnorm_kernel<32>, grid, blk, 0, st, (const bf16_t*)x, (const bf16_t*)gamma, (bf16_t*)y, rows, (int)C, (int)Cpad, apply_silu);
  else
    hipLaunchKernelGGL(vae_rmsnorm_kernel<64>, grid, blk, 0, st, (const bf16_t*)x, (const bf16_t*)gamma, (bf16_t*)y, rows, (int)C, (int)Cpad, apply_silu);
  LCV_LAUNCH_CHECK("vae_rmsnorm_silu");
  return LCV_OK;
}

// p[row, :] = softmax(scale * s[row, :]) ; s fp32 [rows, n] (ld_s), p bf16 [rows, ld_p]; one workgroup per row.
__global__ __launch_bounds__(256) void softmax_rows_kernel(const float* __restrict__ s, bf16_t* __restrict__ p,
                                                           int64_t n, int64_t ld_s, int64_t ld_p, float scale) {
  __shared__ float red[4];
  const int64_t row = blockIdx.x;
  const float* sr = s + row * ld_s;
  bf16_t* pr = p + row * ld_p;
  float mx = -INFINITY;
  for (int64_t i = threadIdx.x; i < n; i += 256) mx = fmaxf(mx, sr[i]);
  mx = wave_max(mx);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = mx;
  __syncthreads();
  mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
  __syncthreads();
  float sum = 0.f;
  for (int64_t i = threadIdx.x; i < n; i += 256) sum += __expf((sr[i] - mx) * scale);
  sum = wave_sum(sum);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = sum;
  __syncthreads();
  const float inv = 1.0f / (red[0] + red[1] + red[2] + red[3]);
  for (int64_t i = threadIdx.x; i < ld_p; i += 256)
    pr[i] = (i < n) ? f2bf(__expf((sr[i] - mx) * scale) * inv) : (bf16_t)0;
}

extern "C" int lcv_softmax_rows(const float* s, void* p, int64_t rows, int64_t n, int64_t ld_s, int64_t ld_p,
                                float scale, void* stream) {
  LCV_CHECK_ARG(s && p && n > 0 && ld_s >= n && ld_p >= n, "softmax_rows: bad arguments");
  if (rows == 0) return LCV_OK;
  hipLaunchKernelGGL(softmax_rows_kernel, dim3((unsigned)rows), dim3(256), 0, (hipStream_t)stream, s, (bf16_t*)p, n,
                     ld_s, ld_p, scale);
  LCV_LAUNCH_CHECK("softmax_rows");
  return LCV_OK;
}
