// UMT5 text encoder pieces that are not a plain GEMM (the reference calls
// transformers.UMT5EncoderModel once per prompt: delta_experiment/scripts/common.py:62-64, 228-255).
//   lcv_gather_rows      token embedding lookup
//   lcv_t5_rmsnorm       T5LayerNorm: y = bf16(w * bf16(x * rsqrt(mean(x^2) + eps))), fp32 statistics, no mean, no bias
//   lcv_geglu_tanh_fwd   gated GELU (tanh form): out = bf16(bf16(gelu_new(gate)) * up)
//   lcv_t5_attention     softmax(q.k^T + relative-position bias + padding mask) . v, d_kv = 64, NO 1/sqrt(d) scaling
// The prompt is at most 512 tokens, so none of this is on MFMA: the encoder's time is in its GEMMs (lcv_gemm_nt); the
// attention below keeps one head's K and V in LDS and walks the queries on the vector pipe.  Rounding points follow the
// bf16 model: scores, scores+bias, probabilities and the output are each rounded to bf16, the softmax itself is fp32.
#include "lcv_common.h"

namespace {

__global__ __launch_bounds__(256) void gather_rows_kernel(const bf16_t* __restrict__ table, const int64_t* __restrict__ ids,
                                                          bf16_t* __restrict__ out, int64_t n, int cpk /*C/8*/, int64_t vocab) {
  const int64_t total = n * cpk;
  for (int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x; p < total; p += (int64_t)gridDim.x * 256) {
    const int64_t row = p / cpk;
    const int c = (int)(p - row * cpk) * 8;
    int64_t id = ids[row];
    id = id < 0 ? 0 : (id >= vocab ? vocab - 1 : id);   // ids are validated on the host; never read out of the table
    *reinterpret_cast<u16x8*>(out + row * (int64_t)cpk * 8 + c) =
        *reinterpret_cast<const u16x8*>(table + id * (int64_t)cpk * 8 + c);
  }
}

constexpr int T5_MAXCH = 8;  // C <= 4096: 8 packets of 8 per lane
__global__ __launch_bounds__(256) void t5_rmsnorm_kernel(const bf16_t* __restrict__ x, const bf16_t* __restrict__ w,
                                                         bf16_t* __restrict__ y, int64_t rows, int C, float eps) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const bf16_t* xr = x + row * C;
  float v[T5_MAXCH][8];
  float sq = 0.f;
#pragma unroll
  for (int ch = 0; ch < T5_MAXCH; ++ch) {
    const int c = (ch * 64 + lane) * 8;
    if (c < C) {
      unpack8(*reinterpret_cast<const u16x8*>(xr + c), v[ch]);
#pragma unroll
      for (int i = 0; i < 8; ++i) sq = fmaf(v[ch][i], v[ch][i], sq);
    }
  }
  const float rstd = rsqrtf(wave_sum(sq) / (float)C + eps);
#pragma unroll
  for (int ch = 0; ch < T5_MAXCH; ++ch) {
    const int c = (ch * 64 + lane) * 8;
    if (c < C) {
      float wv[8], o[8];
      unpack8(*reinterpret_cast<const u16x8*>(w + c), wv);
#pragma unroll
      for (int i = 0; i < 8; ++i) o[i] = wv[i] * bfround(v[ch][i] * rstd);
      *reinterpret_cast<u16x8*>(y + row * C + c) = pack8(o);
    }
  }
}

__device__ __forceinline__ float gelu_new_f(float x) {
  const float u = 0.7978845608028654f * (x + 0.044715f * x * x * x);
  return 0.5f * x * (1.0f + tanhf(u));
}

__global__ __launch_bounds__(256) void geglu_tanh_fwd_kernel(const bf16_t* __restrict__ gate, const bf16_t* __restrict__ up,
                                                             bf16_t* __restrict__ out, int64_t rows, int fpk, int64_t ld_in) {
  const int64_t n_packets = rows * fpk;
  for (int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x; p < n_packets; p += (int64_t)gridDim.x * 256) {
    const int64_t row = p / fpk;
    const int c = (int)(p - row * fpk) * 8;
    float g[8], u[8], o[8];
    unpack8(*reinterpret_cast<const u16x8*>(gate + row * ld_in + c), g);
    unpack8(*reinterpret_cast<const u16x8*>(up + row * ld_in + c), u);
#pragma unroll
    for (int i = 0; i < 8; ++i) o[i] = bfround(gelu_new_f(g[i])) * u[i];
    *reinterpret_cast<u16x8*>(out + p * 8) = pack8(o);
  }
}

// ---------------------------------------------------------------------------
// Attention for one (batch, head, 64-query tile).  LDS: K and V of the head, rows padded to 33 dwords (lane j reads row
// j: an odd dword stride keeps the 64 lanes on distinct banks), plus one fp32 probability row per wave.
// Scores phase: lane l owns keys l, l+64, ...; output phase: lane l owns output channel l.
// ---------------------------------------------------------------------------
constexpr int T5_DK = 64;
constexpr int T5_KROW = 33;  // dwords per K / V row in LDS
constexpr int T5_MAXM = 8;   // S <= 512

__global__ __launch_bounds__(256) void t5_attention_kernel(const bf16_t* __restrict__ q, const bf16_t* __restrict__ k,
                                                           const bf16_t* __restrict__ v, bf16_t* __restrict__ out,
                                                           const float* __restrict__ bias_by_dist /*[H, 2S-1]*/,
                                                           const int* __restrict__ key_mask /*[B, S]*/, int S, int Spad,
                                                           int64_t ld_qkv, int64_t ld_o, int64_t bs_qkv, int64_t bs_o) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned int* Ks = reinterpret_cast<unsigned int*>(smem);
  unsigned int* Vs = Ks + (size_t)Spad * T5_KROW;
  float* Ps = reinterpret_cast<float*>(Vs + (size_t)Spad * T5_KROW);   // [4 waves][Spad]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int h = blockIdx.y, b = blockIdx.z, H = gridDim.y;
  const bf16_t* qb = q + b * bs_qkv + h * T5_DK;
  const bf16_t* kb = k + b * bs_qkv + h * T5_DK;
  const bf16_t* vb = v + b * bs_qkv + h * T5_DK;
  // stage K, V: 32 dwords per row; rows past S are zero
  for (int i = tid; i < Spad * 32; i += 256) {
    const int j = i >> 5, d2 = i & 31;
    unsigned int kk = 0, vv = 0;
    if (j < S) {
      kk = *reinterpret_cast<const unsigned int*>(kb + (int64_t)j * ld_qkv + d2 * 2);
      vv = *reinterpret_cast<const unsigned int*>(vb + (int64_t)j * ld_qkv + d2 * 2);
    }
    Ks[j * T5_KROW + d2] = kk;
    Vs[j * T5_KROW + d2] = vv;
  }
  __syncthreads();
  const int nm = Spad >> 6;
  const float* bias_h = bias_by_dist + (int64_t)h * (2 * S - 1) + (S - 1);   // index by (key - query)
  const int* mk = key_mask + (int64_t)b * S;
  float* Pw = Ps + wave * Spad;
  const unsigned short* Vh = reinterpret_cast<const unsigned short*>(Vs);
  for (int qi = 0; qi < 16; ++qi) {
    const int i = blockIdx.x * 64 + wave * 16 + qi;      // wave-uniform
    if (i >= S) break;
    // the query row, the same 32 dwords in every lane
    unsigned int qr[32];
#pragma unroll
    for (int d4 = 0; d4 < 8; ++d4) {
      const u32x4 t = *reinterpret_cast<const u32x4*>(qb + (int64_t)i * ld_qkv + d4 * 8);
      qr[d4 * 4 + 0] = t[0]; qr[d4 * 4 + 1] = t[1]; qr[d4 * 4 + 2] = t[2]; qr[d4 * 4 + 3] = t[3];
    }
    float sc[T5_MAXM];
    float mx = -INFINITY;
#pragma unroll
    for (int m = 0; m < T5_MAXM; ++m) {
      sc[m] = -INFINITY;
      if (m < nm) {
        const int j = lane + 64 * m;
        float acc = 0.f;
#pragma unroll
        for (int d2 = 0; d2 < 32; ++d2) {
          const unsigned int kk = Ks[j * T5_KROW + d2], qq = qr[d2];
          acc = fmaf(__builtin_bit_cast(float, qq << 16), __builtin_bit_cast(float, kk << 16), acc);
          acc = fmaf(__builtin_bit_cast(float, qq & 0xffff0000u), __builtin_bit_cast(float, kk & 0xffff0000u), acc);
        }
        if (j < S && mk[j] != 0) sc[m] = bfround(bfround(acc) + bias_h[j - i]);
        mx = fmaxf(mx, sc[m]);
      }
    }
    mx = wave_max(mx);
    float sum = 0.f;
#pragma unroll
    for (int m = 0; m < T5_MAXM; ++m) {
      if (m < nm) {
        sc[m] = (mx == -INFINITY) ? 0.f : __expf(sc[m] - mx);
        sum += sc[m];
      }
    }
    sum = wave_sum(sum);
    const float inv = sum > 0.f ? 1.0f / sum : 0.f;
#pragma unroll
    for (int m = 0; m < T5_MAXM; ++m)
      if (m < nm) Pw[lane + 64 * m] = bfround(sc[m] * inv);
    // the probabilities were written by this wave only: no workgroup barrier, just LDS visibility within the wave
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
    float o = 0.f;
    for (int j = 0; j < Spad; j += 4) {
      const f32x4 pj = *reinterpret_cast<const f32x4*>(Pw + j);
#pragma unroll
      for (int u = 0; u < 4; ++u) o = fmaf(pj[u], bf2f(Vh[(size_t)(j + u) * (T5_KROW * 2) + lane]), o);
    }
    out[b * bs_o + (int64_t)i * ld_o + h * T5_DK + lane] = f2bf(o);
    __builtin_amdgcn_wave_barrier();      // Pw is rewritten by the next query
  }
  (void)H;
}

}  // namespace

extern "C" int lcv_gather_rows(const void* table, const int64_t* ids, void* out, int64_t n, int64_t C, int64_t vocab,
                               void* stream) {
  LCV_CHECK_ARG(table && ids && out, "gather_rows: null pointer");
  LCV_CHECK_ARG(C > 0 && C % 8 == 0 && vocab > 0, "gather_rows: C=%ld must be a multiple of 8", (long)C);
  if (n == 0) return LCV_OK;
  int64_t blocks = (n * (C / 8) + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(gather_rows_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)table,
                     ids, (bf16_t*)out, n, (int)(C / 8), vocab);
  LCV_LAUNCH_CHECK("gather_rows");
  return LCV_OK;
}

extern "C" int lcv_t5_rmsnorm(const void* x, const void* w, void* y, int64_t rows, int64_t C, float eps, void* stream) {
  LCV_CHECK_ARG(x && w && y, "t5_rmsnorm: null pointer");
  LCV_CHECK_ARG(C > 0 && C % 8 == 0 && C <= 4096, "t5_rmsnorm: C=%ld must be a multiple of 8 and <= 4096", (long)C);
  if (rows == 0) return LCV_OK;
  hipLaunchKernelGGL(t5_rmsnorm_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, (hipStream_t)stream,
                     (const bf16_t*)x, (const bf16_t*)w, (bf16_t*)y, rows, (int)C, eps);
  LCV_LAUNCH_CHECK("t5_rmsnorm");
  return LCV_OK;
}

extern "C" int lcv_geglu_tanh_fwd(const void* gate, const void* up, void* out, int64_t rows, int64_t F, int64_t ld_in,
                                  void* stream) {
  LCV_CHECK_ARG(gate && up && out, "geglu_tanh_fwd: null pointer");
  LCV_CHECK_ARG(F % 8 == 0 && ld_in % 8 == 0, "geglu_tanh_fwd: F and ld_in must be multiples of 8");
  const int64_t n_packets = rows * (F / 8);
  if (n_packets == 0) return LCV_OK;
  int64_t blocks = (n_packets + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(geglu_tanh_fwd_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)gate,
                     (const bf16_t*)up, (bf16_t*)out, rows, (int)(F / 8), ld_in);
  LCV_LAUNCH_CHECK("geglu_tanh_fwd");
  return LCV_OK;
}

extern "C" int lcv_t5_attention(const void* q, const void* k, const void* v, void* out, const float* bias_by_dist,
                                const int* key_mask, int64_t B, int64_t S, int64_t H, int64_t ld_qkv, int64_t ld_o,
                                int64_t bs_qkv, int64_t bs_o, void* stream) {
  LCV_CHECK_ARG(q && k && v && out && bias_by_dist && key_mask, "t5_attention: null pointer");
  LCV_CHECK_ARG(S > 0 && S <= 64 * T5_MAXM, "t5_attention: S=%ld tokens (1..%d)", (long)S, 64 * T5_MAXM);
  LCV_CHECK_ARG(B > 0 && B <= 65535 && H > 0 && H <= 65535, "t5_attention: bad batch / head count");
  LCV_CHECK_ARG(ld_qkv % 8 == 0 && ld_qkv >= T5_DK && ld_o >= H * T5_DK, "t5_attention: row strides (d_kv is fixed at %d)", T5_DK);
  const int Spad = (int)((S + 63) / 64 * 64);
  const size_t lds = (size_t)Spad * T5_KROW * 4 * 2 + (size_t)4 * Spad * 4;
  static size_t lds_set = 0;
  if (lds > lds_set) {
    if (hipFuncSetAttribute((const void*)t5_attention_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) {
      lcv_set_error("t5_attention: cannot reserve %zu bytes of LDS", lds);
      return LCV_ELAUNCH;
    }
    lds_set = lds;
  }
  hipLaunchKernelGGL(t5_attention_kernel, dim3((unsigned)(Spad / 64), (unsigned)H, (unsigned)B), dim3(256), lds,
                     (hipStream_t)stream, (const bf16_t*)q, (const bf16_t*)k, (const bf16_t*)v, (bf16_t*)out, bias_by_dist,
                     key_mask, (int)S, Spad, ld_qkv, ld_o, bs_qkv, bs_o);
  LCV_LAUNCH_CHECK("t5_attention");
  return LCV_OK;
}
