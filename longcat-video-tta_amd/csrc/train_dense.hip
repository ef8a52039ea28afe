// Backward pieces that only full-model TTA needs (lora_experiment/scripts/run_full_tta.py:95-215: every DiT parameter
// trainable, SGD or AdamW): dense weight gradients, bias gradients, the fp32-island linears' weight gradients and the
// caption embedder's GELU.
//
// Dense dW[N,K] = dY^T[N,M] . X[M,K] has the token axis M as its contraction; the MFMA GEMM of this library contracts
// over the contiguous axis ("NT"), so both operands are first written transposed and zero-padded to a 64-multiple of
// tokens (lcv_transpose_pad, an HBM-bound pass: 4 B per element moved) and dW is then ONE lcv_gemm_nt(dY^T, X^T) on the
// 8-phase kernel.  The transposed dY also gives the bias gradient as row sums (lcv_rowsum).
#include "lcv_common.h"

namespace {

// in [M, N] (row stride ld) -> out [N, Mpad]; columns m >= M are zero.  64x64 tiles through LDS, 8-byte accesses.
__global__ __launch_bounds__(256) void transpose_pad_kernel(const bf16_t* __restrict__ in, bf16_t* __restrict__ out,
                                                            int64_t M, int64_t N, int64_t ld, int64_t Mpad) {
  __shared__ bf16_t tile[64][68];   // [m][n], padded: the transposed read walks a column
  const int tid = threadIdx.x;
  const int64_t m0 = (int64_t)blockIdx.x * 64, n0 = (int64_t)blockIdx.y * 64;
#pragma unroll
  for (int it = 0; it < 4; ++it) {
    const int idx = tid + it * 256;       // 1024 packets of 4
    const int r = idx >> 4, c4 = (idx & 15) * 4;
    u16x4 v = {0, 0, 0, 0};
    const int64_t m = m0 + r, n = n0 + c4;
    if (m < M) {
      if (n + 3 < N && ((ld | n) & 3) == 0) {
        v = *reinterpret_cast<const u16x4*>(in + m * ld + n);
      } else {
#pragma unroll
        for (int e = 0; e < 4; ++e)
          if (n + e < N) v[e] = in[m * ld + n + e];
      }
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) tile[r][c4 + e] = v[e];
  }
  __syncthreads();
#pragma unroll
  for (int it = 0; it < 4; ++it) {
    const int idx = tid + it * 256;
    const int r = idx >> 4, c4 = (idx & 15) * 4;   // r: n within tile, c4: m within tile
    const int64_t n = n0 + r, m = m0 + c4;
    if (n < N && m < Mpad) {                        // Mpad % 64 == 0: whole packets
      u16x4 v;
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = tile[c4 + e][r];
      *reinterpret_cast<u16x4*>(out + n * Mpad + m) = v;
    }
  }
}

// out[r] = sum_c in[r, c] (fp32 accumulate), one wave per row; out bf16 or fp32
__global__ __launch_bounds__(256) void rowsum_kernel(const bf16_t* __restrict__ in, void* __restrict__ out, int64_t rows,
                                                     int64_t cols, int out_f32) {
  const int lane = threadIdx.x & 63;
  const int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= rows) return;
  float acc = 0.f;
  for (int64_t c = lane * 8; c < cols; c += 64 * 8) {   // cols % 8 == 0
    float v[8];
    unpack8(*reinterpret_cast<const u16x8*>(in + r * cols + c), v);
#pragma unroll
    for (int i = 0; i < 8; ++i) acc += v[i];
  }
  acc = wave_sum(acc);
  if (lane == 0) {
    if (out_f32) ((float*)out)[r] = acc;
    else ((bf16_t*)out)[r] = f2bf(acc);
  }
}

__device__ __forceinline__ float silu_f32(float x) { return x / (1.0f + __expf(-x)); }

// fp32-island linear y = act(a) w^T + b with M <= 64 rows: dw[n,k] = sum_m dy[m,n] act(a[m,k]), db[n] = sum_m dy[m,n]
__global__ __launch_bounds__(256) void smallm_wgrad_kernel(const float* __restrict__ dy, const float* __restrict__ a,
                                                           bf16_t* __restrict__ dw, bf16_t* __restrict__ db, int M, int64_t N,
                                                           int64_t K, int act_in) {
  const int64_t n = blockIdx.y;
  const int64_t k = (int64_t)blockIdx.x * 256 + threadIdx.x;
  float acc = 0.f, bsum = 0.f;
  for (int m = 0; m < M; ++m) {
    const float g = dy[(int64_t)m * N + n];
    bsum += g;
    if (k < K) {
      float av = a[(int64_t)m * K + k];
      if (act_in == 1) av = silu_f32(av);
      acc = fmaf(g, av, acc);
    }
  }
  if (k < K) dw[n * K + k] = f2bf(acc);
  if (db && blockIdx.x == 0 && threadIdx.x == 0) db[n] = f2bf(bsum);
}

__device__ __forceinline__ float gelu_tanh_f(float x) {
  const float u = 0.7978845608028654f * (x + 0.044715f * x * x * x);
  return 0.5f * x * (1.0f + tanhf(u));
}
__device__ __forceinline__ float gelu_tanh_grad_f(float x) {
  const float u = 0.7978845608028654f * (x + 0.044715f * x * x * x);
  const float t = tanhf(u);
  const float du = 0.7978845608028654f * (1.0f + 3.0f * 0.044715f * x * x);
  return 0.5f * (1.0f + t) + 0.5f * x * (1.0f - t * t) * du;
}

template <bool BWD>
__global__ __launch_bounds__(256) void gelu_tanh_kernel(const bf16_t* __restrict__ x, const bf16_t* __restrict__ dy,
                                                        bf16_t* __restrict__ out, int64_t n_packets) {
  for (int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x; p < n_packets; p += (int64_t)gridDim.x * 256) {
    float xv[8], o[8];
    unpack8(*reinterpret_cast<const u16x8*>(x + p * 8), xv);
    if (BWD) {
      float g[8];
      unpack8(*reinterpret_cast<const u16x8*>(dy + p * 8), g);
#pragma unroll
      for (int i = 0; i < 8; ++i) o[i] = g[i] * gelu_tanh_grad_f(xv[i]);
    } else {
#pragma unroll
      for (int i = 0; i < 8; ++i) o[i] = gelu_tanh_f(xv[i]);
    }
    *reinterpret_cast<u16x8*>(out + p * 8) = pack8(o);
  }
}

}  // namespace

extern "C" int lcv_transpose_pad(const void* in, void* out, int64_t M, int64_t N, int64_t ld, int64_t Mpad, void* stream) {
  LCV_CHECK_ARG(in && out, "transpose_pad: null pointer");
  LCV_CHECK_ARG(M > 0 && N > 0 && ld >= N && Mpad >= M && Mpad % 64 == 0, "transpose_pad: M=%ld N=%ld ld=%ld Mpad=%ld (Mpad must be a multiple of 64)",
                (long)M, (long)N, (long)ld, (long)Mpad);
  LCV_CHECK_ARG(Mpad / 64 <= 0x7fffffff && (N + 63) / 64 <= 65535, "transpose_pad: grid too large");
  hipLaunchKernelGGL(transpose_pad_kernel, dim3((unsigned)(Mpad / 64), (unsigned)((N + 63) / 64)), dim3(256), 0,
                     (hipStream_t)stream, (const bf16_t*)in, (bf16_t*)out, M, N, ld, Mpad);
  LCV_LAUNCH_CHECK("transpose_pad");
  return LCV_OK;
}

extern "C" int lcv_rowsum(const void* in, void* out, int64_t rows, int64_t cols, int out_f32, void* stream) {
  LCV_CHECK_ARG(in && out && rows > 0 && cols > 0 && cols % 8 == 0, "rowsum: bad arguments (cols must be a multiple of 8)");
  hipLaunchKernelGGL(rowsum_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)in, out,
                     rows, cols, out_f32);
  LCV_LAUNCH_CHECK("rowsum");
  return LCV_OK;
}

extern "C" int lcv_linear_f32_smallm_wgrad(const float* dy, const float* a, void* dw, void* db, int64_t M, int64_t N, int64_t K,
                                           int act_in, void* stream) {
  LCV_CHECK_ARG(dy && a && dw, "linear_f32_smallm_wgrad: null pointer");
  LCV_CHECK_ARG(M > 0 && M <= 4096 && N > 0 && N <= 65535 && K > 0, "linear_f32_smallm_wgrad: M=%ld N=%ld K=%ld", (long)M, (long)N, (long)K);
  hipLaunchKernelGGL(smallm_wgrad_kernel, dim3((unsigned)((K + 255) / 256), (unsigned)N), dim3(256), 0, (hipStream_t)stream, dy, a,
                     (bf16_t*)dw, (bf16_t*)db, (int)M, N, K, act_in);
  LCV_LAUNCH_CHECK("linear_f32_smallm_wgrad");
  return LCV_OK;
}

extern "C" int lcv_gelu_tanh_fwd(const void* x, void* y, int64_t n, void* stream) {
  LCV_CHECK_ARG(x && y && n >= 0 && n % 8 == 0, "gelu_tanh_fwd: n must be a multiple of 8");
  if (n == 0) return LCV_OK;
  int64_t blocks = (n / 8 + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(gelu_tanh_kernel<false>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)x,
                     (const bf16_t*)nullptr, (bf16_t*)y, n / 8);
  LCV_LAUNCH_CHECK("gelu_tanh_fwd");
  return LCV_OK;
}

extern "C" int lcv_gelu_tanh_bwd(const void* x, const void* dy, void* dx, int64_t n, void* stream) {
  LCV_CHECK_ARG(x && dy && dx && n >= 0 && n % 8 == 0, "gelu_tanh_bwd: n must be a multiple of 8");
  if (n == 0) return LCV_OK;
  int64_t blocks = (n / 8 + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(gelu_tanh_kernel<true>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)x,
                     (const bf16_t*)dy, (bf16_t*)dx, n / 8);
  LCV_LAUNCH_CHECK("gelu_tanh_bwd");
  return LCV_OK;
}
