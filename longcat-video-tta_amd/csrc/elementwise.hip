// HBM-bound forward kernels of the DiT block: AdaLN modulate / affine LayerNorm,
// gated residual, q/k RMSNorm + 3-D RoPE, SwiGLU, patchify/unpatchify, and the
// fp32 denoise-step / flow-matching-loss glue.  Every kernel moves 16 B per lane
// per access (8 bf16) and keeps its statistics in fp32.
#include "lcv_common.h"

// ---------------------------------------------------------------------------
// Row LayerNorm (fp32 statistics), one wave per row, row kept in registers.
//   MODE 0: y = xhat * (1 + scale[frame]) + shift[frame]   (AdaLN modulate)
//   MODE 1: y = xhat * w + b                                (affine LayerNorm_FP32)
// Algorithmic bytes per row: 2*C*2 (+ 2*C*4 of parameters, L2-resident).
// ---------------------------------------------------------------------------
#define ROWNORM_MAXCH 8  // 8 chunks * 64 lanes * 8 elems = 4096 channels max

template <int MODE>
__global__ __launch_bounds__(256) void rownorm_fwd_kernel(
    const bf16_t* __restrict__ x, const float* __restrict__ p_add, const float* __restrict__ p_mul,
    bf16_t* __restrict__ y, int64_t rows, int C, int64_t S, int64_t mod_stride, float eps) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const bf16_t* xr = x + row * C;
  float v[ROWNORM_MAXCH][8];
  float sum = 0.f;
#pragma unroll
  for (int ch = 0; ch < ROWNORM_MAXCH; ++ch) {
    const int c = (ch * 64 + lane) * 8;
    if (c < C) {
      u16x8 raw = *reinterpret_cast<const u16x8*>(xr + c);
      unpack8(raw, v[ch]);
#pragma unroll
      for (int i = 0; i < 8; ++i) sum += v[ch][i];
    } else {
#pragma unroll
      for (int i = 0; i < 8; ++i) v[ch][i] = 0.f;
    }
  }
  const float mean = wave_sum(sum) / (float)C;
  float sq = 0.f;
#pragma unroll
  for (int ch = 0; ch < ROWNORM_MAXCH; ++ch) {
    const int c = (ch * 64 + lane) * 8;
    if (c < C) {
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const float d = v[ch][i] - mean;
        sq += d * d;
      }
    }
  }
  const float rstd = rsqrtf(wave_sum(sq) / (float)C + eps);
  const float* pa = p_add;
  const float* pm = p_mul;
  if (MODE == 0) {
    const int64_t frame = row / S;
    pa += frame * mod_stride;
    pm += frame * mod_stride;
  }
  bf16_t* yr = y + row * C;
#pragma unroll
  for (int ch = 0; ch < ROWNORM_MAXCH; ++ch) {
    const int c = (ch * 64 + lane) * 8;
    if (c < C) {
      const f32x4 a0 = *reinterpret_cast<const f32x4*>(pa + c);
      const f32x4 a1 = *reinterpret_cast<const f32x4*>(pa + c + 4);
      const f32x4 m0 = *reinterpret_cast<const f32x4*>(pm + c);
      const f32x4 m1 = *reinterpret_cast<const f32x4*>(pm + c + 4);
      float o[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const float xh = (v[ch][i] - mean) * rstd;
        const float mul = (i < 4) ? m0[i] : m1[i - 4];
        const float add = (i < 4) ? a0[i] : a1[i - 4];
        o[i] = (MODE == 0) ? (xh * (mul + 1.0f) + add) : (xh * mul + add);
      }
      *reinterpret_cast<u16x8*>(yr + c) = pack8(o);
    }
  }
}

extern "C" int lcv_adaln_modulate_fwd(const void* x, const float* mod, void* y, int64_t B, int64_t T,
                                      int64_t S, int64_t C, int64_t mod_stride, int64_t shift_off,
                                      int64_t scale_off, float eps, void* stream) {
  LCV_CHECK_ARG(x && mod && y, "adaln_modulate_fwd: null pointer");
  LCV_CHECK_ARG(C > 0 && C % 8 == 0 && C <= 4096, "adaln_modulate_fwd: C=%ld must be a multiple of 8 and <= 4096", (long)C);
  LCV_CHECK_ARG(B > 0 && T > 0 && S > 0, "adaln_modulate_fwd: bad sizes");
  LCV_CHECK_ARG(shift_off % 4 == 0 && scale_off % 4 == 0 && mod_stride % 4 == 0, "adaln_modulate_fwd: offsets must be 16-byte aligned");
  const int64_t rows = B * T * S;
  const unsigned grid = (unsigned)((rows + 3) / 4);
  hipLaunchKernelGGL(rownorm_fwd_kernel<0>, dim3(grid), dim3(256), 0, (hipStream_t)stream,
                     (const bf16_t*)x, mod + shift_off, mod + scale_off, (bf16_t*)y, rows, (int)C, S,
                     mod_stride, eps);
  LCV_LAUNCH_CHECK("adaln_modulate_fwd");
  return LCV_OK;
}

extern "C" int lcv_layernorm_affine_fwd(const void* x, const float* w, const float* b, void* y,
                                        int64_t rows, int64_t C, float eps, void* stream) {
  LCV_CHECK_ARG(x && w && b && y, "layernorm_affine_fwd: null pointer");
  LCV_CHECK_ARG(C > 0 && C % 8 == 0 && C <= 4096, "layernorm_affine_fwd: C=%ld must be a multiple of 8 and <= 4096", (long)C);
  if (rows == 0) return LCV_OK;
  const unsigned grid = (unsigned)((rows + 3) / 4);
  hipLaunchKernelGGL(rownorm_fwd_kernel<1>, dim3(grid), dim3(256), 0, (hipStream_t)stream,
                     (const bf16_t*)x, b, w, (bf16_t*)y, rows, (int)C, (int64_t)1, (int64_t)0, eps);
  LCV_LAUNCH_CHECK("layernorm_affine_fwd");
  return LCV_OK;
}

// ---------------------------------------------------------------------------
// Gated residual: out = bf16(f32(x) + gate[frame] * f32(y)).  3 streams of
// N*C*2 bytes; grid-stride over 16-byte packets.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void gate_residual_fwd_kernel(
    const bf16_t* __restrict__ x, const bf16_t* __restrict__ y, const float* __restrict__ gate,
    bf16_t* __restrict__ out, int64_t n_packets, int cpk /*C/8*/, int64_t S, int64_t mod_stride) {
  for (int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; p < n_packets;
       p += (int64_t)gridDim.x * blockDim.x) {
    const int64_t row = p / cpk;
    const int c = (int)(p - row * cpk) * 8;
    float xf[8], yf[8], o[8];
    unpack8(*reinterpret_cast<const u16x8*>(x + p * 8), xf);
    unpack8(*reinterpret_cast<const u16x8*>(y + p * 8), yf);
    if (gate) {
      const float* g = gate + (row / S) * mod_stride + c;
      const f32x4 g0 = *reinterpret_cast<const f32x4*>(g);
      const f32x4 g1 = *reinterpret_cast<const f32x4*>(g + 4);
#pragma unroll
      for (int i = 0; i < 8; ++i) o[i] = xf[i] + ((i < 4) ? g0[i] : g1[i - 4]) * yf[i];
    } else {
#pragma unroll
      for (int i = 0; i < 8; ++i) o[i] = xf[i] + yf[i];
    }
    *reinterpret_cast<u16x8*>(out + p * 8) = pack8(o);
  }
}

extern "C" int lcv_gate_residual_fwd(const void* x, const void* y, const float* mod, void* out,
                                     int64_t B, int64_t T, int64_t S, int64_t C, int64_t mod_stride,
                                     int64_t gate_off, void* stream) {
  LCV_CHECK_ARG(x && y && out, "gate_residual_fwd: null pointer");
  LCV_CHECK_ARG(C > 0 && C % 8 == 0, "gate_residual_fwd: C must be a multiple of 8");
  LCV_CHECK_ARG(gate_off % 4 == 0 && mod_stride % 4 == 0, "gate_residual_fwd: offsets must be 16-byte aligned");
  const int64_t n_packets = B * T * S * (C / 8);
  if (n_packets == 0) return LCV_OK;
  int64_t blocks = (n_packets + 255) / 256;
  if (blocks > 256 * 16) blocks = 256 * 16;
  hipLaunchKernelGGL(gate_residual_fwd_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream,
                     (const bf16_t*)x, (const bf16_t*)y, mod ? mod + gate_off : nullptr, (bf16_t*)out,
                     n_packets, (int)(C / 8), S, mod_stride);
  LCV_LAUNCH_CHECK("gate_residual_fwd");
  return LCV_OK;
}

// ---------------------------------------------------------------------------
// q/k RMSNorm over D=128 (fp32 stats, bf16 roundings of the upstream module
// chain) + interleaved-pair 3-D RoPE from a (cos,sin) table.  One workgroup
// per token; 16 lanes per head (8 elements each), 16 heads per pass, so the
// token's 64 (cos,sin) pairs are loaded once and reused by every head of q and k.
// ---------------------------------------------------------------------------
__device__ __forceinline__ void norm_rope_vec(const bf16_t* src, bf16_t* dst, const float (&w)[8],
                                              const float (&cs)[8], bool do_rope, float eps, float out_scale) {
  float f[8];
  unpack8(*reinterpret_cast<const u16x8*>(src), f);
  float ss = 0.f;
#pragma unroll
  for (int i = 0; i < 8; ++i) ss += f[i] * f[i];
  // reduce across the 16 lanes of this head
  ss += __shfl_xor(ss, 8, 64);
  ss += __shfl_xor(ss, 4, 64);
  ss += __shfl_xor(ss, 2, 64);
  ss += __shfl_xor(ss, 1, 64);
  const float r = rsqrtf(ss * (1.0f / 128.0f) + eps);
  float o[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) f[i] = bfround(bfround(f[i] * r) * w[i]);  // .type_as(x) then * weight (bf16)
  if (do_rope) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const float c = cs[2 * i], s = cs[2 * i + 1];
      const float x0 = f[2 * i], x1 = f[2 * i + 1];
      // q*cos + rotate_half(q)*sin, each product rounded in fp32 as torch does
      o[2 * i] = __fadd_rn(__fmul_rn(x0, c), __fmul_rn(-x1, s));
      o[2 * i + 1] = __fadd_rn(__fmul_rn(x1, c), __fmul_rn(x0, s));
    }
  } else {
#pragma unroll
    for (int i = 0; i < 8; ++i) o[i] = f[i];
  }
  // out_scale != 1 (q only): the attention scale head_dim^-0.5 * log2(e) folded in BEFORE the one bf16 rounding of the
  // output, so the attention kernel's exponent is q.k itself (attn_fwd.hip, UNIT)
#pragma unroll
  for (int i = 0; i < 8; ++i) o[i] *= out_scale;
  *reinterpret_cast<u16x8*>(dst) = pack8(o);
}

__global__ __launch_bounds__(256) void qknorm_rope_fwd_kernel(
    const bf16_t* __restrict__ q_in, const bf16_t* __restrict__ k_in, const bf16_t* __restrict__ v_in,
    bf16_t* __restrict__ q_out, bf16_t* __restrict__ k_out, bf16_t* __restrict__ v_out,
    const bf16_t* __restrict__ wq, const bf16_t* __restrict__ wk, const float* __restrict__ cs_tab,
    int H, int64_t in_sb, int64_t in_sn, int64_t q_sb, int64_t q_sn, int64_t kv_sb, int64_t kv_sn,
    int64_t pos_off, float eps, float q_scale) {
  const int64_t n = blockIdx.x, b = blockIdx.y;
  const int sub = threadIdx.x & 15;   // 8-element slice of the head vector
  const int hl = threadIdx.x >> 4;    // head within the pass
  float cs[8] = {1, 0, 1, 0, 1, 0, 1, 0};
  const bool do_rope = cs_tab != nullptr;
  if (do_rope) {
    const float* p = cs_tab + ((pos_off + n) * 64 + sub * 4) * 2;
    const f32x4 a = *reinterpret_cast<const f32x4*>(p);
    const f32x4 c = *reinterpret_cast<const f32x4*>(p + 4);
    cs[0] = a[0]; cs[1] = a[1]; cs[2] = a[2]; cs[3] = a[3];
    cs[4] = c[0]; cs[5] = c[1]; cs[6] = c[2]; cs[7] = c[3];
  }
  float wqf[8], wkf[8];
  unpack8(*reinterpret_cast<const u16x8*>(wq + sub * 8), wqf);
  unpack8(*reinterpret_cast<const u16x8*>(wk + sub * 8), wkf);
  const int64_t in_base = b * in_sb + n * in_sn;
  for (int h0 = 0; h0 < H; h0 += 16) {
    const int h = h0 + hl;
    if (h >= H) continue;  // whole 16-lane groups drop out together, so the in-group shuffles stay valid
    const int64_t off = (int64_t)h * 128 + sub * 8;
    if (q_in) norm_rope_vec(q_in + in_base + off, q_out + b * q_sb + n * q_sn + off, wqf, cs, do_rope, eps, q_scale);
    if (k_in) norm_rope_vec(k_in + in_base + off, k_out + b * kv_sb + n * kv_sn + off, wkf, cs, do_rope, eps, 1.0f);
    if (v_out) {
      *reinterpret_cast<u16x8*>(v_out + b * kv_sb + n * kv_sn + off) =
          *reinterpret_cast<const u16x8*>(v_in + in_base + off);
    }
  }
}

extern "C" int lcv_qknorm_rope_fwd(const void* q_in, const void* k_in, const void* v_in, void* q_out,
                                   void* k_out, void* v_out, const void* wq, const void* wk,
                                   const void* cs, int64_t B, int64_t N, int64_t H, int64_t in_sb,
                                   int64_t in_sn, int64_t q_sb, int64_t q_sn, int64_t kv_sb,
                                   int64_t kv_sn, int64_t pos_off, float eps, float q_scale, void* stream) {
  LCV_CHECK_ARG((q_in || k_in) && wq && wk, "qknorm_rope_fwd: null pointer");
  LCV_CHECK_ARG(q_scale > 0.f, "qknorm_rope_fwd: q_scale must be positive (1 = none)");
  LCV_CHECK_ARG(!q_in || q_out, "qknorm_rope_fwd: q_out missing");
  LCV_CHECK_ARG(!k_in || k_out, "qknorm_rope_fwd: k_out missing");
  LCV_CHECK_ARG(H > 0, "qknorm_rope_fwd: H must be positive");
  LCV_CHECK_ARG(in_sn % 8 == 0 && q_sn % 8 == 0 && kv_sn % 8 == 0 && in_sb % 8 == 0 && q_sb % 8 == 0 && kv_sb % 8 == 0,
                "qknorm_rope_fwd: strides must be multiples of 8 elements");
  if (B == 0 || N == 0) return LCV_OK;
  const bool copy_v = v_out && v_in && v_out != v_in;
  hipLaunchKernelGGL(qknorm_rope_fwd_kernel, dim3((unsigned)N, (unsigned)B), dim3(256), 0,
                     (hipStream_t)stream, (const bf16_t*)q_in, (const bf16_t*)k_in, (const bf16_t*)v_in,
                     (bf16_t*)q_out, (bf16_t*)k_out, copy_v ? (bf16_t*)v_out : nullptr, (const bf16_t*)wq,
                     (const bf16_t*)wk, (const float*)cs, (int)H, in_sb, in_sn, q_sb, q_sn, kv_sb, kv_sn,
                     pos_off, eps, q_scale);
  LCV_LAUNCH_CHECK("qknorm_rope_fwd");
  return LCV_OK;
}

// ---------------------------------------------------------------------------
// SwiGLU: out = bf16( bf16(silu(gate)) * up )
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void swiglu_fwd_kernel(const bf16_t* __restrict__ gate,
                                                         const bf16_t* __restrict__ up,
                                                         bf16_t* __restrict__ out, int64_t rows,
                                                         int fpk /*F/8*/, int64_t ld_in) {
  const int64_t n_packets = rows * fpk;
  for (int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; p < n_packets;
       p += (int64_t)gridDim.x * blockDim.x) {
    const int64_t row = p / fpk;
    const int c = (int)(p - row * fpk) * 8;
    float g[8], u[8], o[8];
    unpack8(*reinterpret_cast<const u16x8*>(gate + row * ld_in + c), g);
    unpack8(*reinterpret_cast<const u16x8*>(up + row * ld_in + c), u);
#pragma unroll
    for (int i = 0; i < 8; ++i) o[i] = bfround(silu_f(g[i])) * u[i];
    *reinterpret_cast<u16x8*>(out + p * 8) = pack8(o);
  }
}

extern "C" int lcv_swiglu_fwd(const void* gate, const void* up, void* out, int64_t rows, int64_t F,
                              int64_t ld_in, void* stream) {
  LCV_CHECK_ARG(gate && up && out, "swiglu_fwd: null pointer");
  LCV_CHECK_ARG(F % 8 == 0 && ld_in % 8 == 0, "swiglu_fwd: F and ld_in must be multiples of 8");
  const int64_t n_packets = rows * (F / 8);
  if (n_packets == 0) return LCV_OK;
  int64_t blocks = (n_packets + 255) / 256;
  if (blocks > 256 * 16) blocks = 256 * 16;
  hipLaunchKernelGGL(swiglu_fwd_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream,
                     (const bf16_t*)gate, (const bf16_t*)up, (bf16_t*)out, rows, (int)(F / 8), ld_in);
  LCV_LAUNCH_CHECK("swiglu_fwd");
  return LCV_OK;
}

// ---------------------------------------------------------------------------
// patchify: x [B,Cin,T,H,W] -> tok [B, T*(H/2)*(W/2), Kpad], k = c*4 + ph*2 + pw
// One thread per (token, channel): reads two 4-byte pairs, writes 8 bytes.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void patchify_kernel(const bf16_t* __restrict__ x,
                                                       bf16_t* __restrict__ tok, int64_t B, int Cin,
                                                       int T, int H, int W, int Kpad) {
  const int Hh = H / 2, Wh = W / 2;
  const int kq = Kpad / 4;  // 4-element groups per token (first Cin are channels, rest zero pad)
  const int64_t total = (int64_t)B * T * Hh * Wh * kq;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (int64_t)gridDim.x * blockDim.x) {
    const int c = (int)(i % kq);
    int64_t tkn = i / kq;
    const int wq = (int)(tkn % Wh);
    int64_t r = tkn / Wh;
    const int hq = (int)(r % Hh);
    r /= Hh;
    const int t = (int)(r % T);
    const int64_t b = r / T;
    u16x4 o = {0, 0, 0, 0};
    if (c < Cin) {
      const bf16_t* p = x + ((((b * Cin + c) * T + t) * H + 2 * hq) * (int64_t)W + 2 * wq);
      const u16x2 r0 = *reinterpret_cast<const u16x2*>(p);
      const u16x2 r1 = *reinterpret_cast<const u16x2*>(p + W);
      o[0] = r0[0]; o[1] = r0[1]; o[2] = r1[0]; o[3] = r1[1];
    }
    *reinterpret_cast<u16x4*>(tok + tkn * Kpad + c * 4) = o;
  }
}

extern "C" int lcv_patchify(const void* x, void* tok, int64_t B, int64_t Cin, int64_t T, int64_t H,
                            int64_t W, int64_t Kpad, void* stream) {
  LCV_CHECK_ARG(x && tok, "patchify: null pointer");
  LCV_CHECK_ARG(H % 2 == 0 && W % 2 == 0, "patchify: H, W must be even (patch 1x2x2)");
  LCV_CHECK_ARG(Kpad % 4 == 0 && Kpad >= Cin * 4, "patchify: Kpad must be a multiple of 4 and >= 4*Cin");
  const int64_t total = B * T * (H / 2) * (W / 2) * (Kpad / 4);
  if (total == 0) return LCV_OK;
  int64_t blocks = (total + 255) / 256;
  if (blocks > 256 * 16) blocks = 256 * 16;
  hipLaunchKernelGGL(patchify_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream,
                     (const bf16_t*)x, (bf16_t*)tok, B, (int)Cin, (int)T, (int)H, (int)W, (int)Kpad);
  LCV_LAUNCH_CHECK("patchify");
  return LCV_OK;
}

// unpatchify: tok [B, N, (ph pw c)] -> out [B,Cout,T,H,W] fp32
template <typename TokT>
__global__ __launch_bounds__(256) void unpatchify_kernel(const TokT* __restrict__ tok,
                                                         float* __restrict__ out, int64_t B, int Cout,
                                                         int T, int H, int W) {
  const int Hh = H / 2, Wh = W / 2;
  const int64_t total = (int64_t)B * Cout * T * H * W;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (int64_t)gridDim.x * blockDim.x) {
    const int w = (int)(i % W);
    int64_t r = i / W;
    const int h = (int)(r % H);
    r /= H;
    const int t = (int)(r % T);
    r /= T;
    const int c = (int)(r % Cout);
    const int64_t b = r / Cout;
    const int64_t tkn = ((b * T + t) * Hh + (h >> 1)) * Wh + (w >> 1);
    const int k = ((h & 1) * 2 + (w & 1)) * Cout + c;
    const TokT vraw = tok[tkn * (4 * Cout) + k];
    float v;
    if constexpr (sizeof(TokT) == 2) v = bf2f((bf16_t)vraw); else v = (float)vraw;
    out[i] = v;
  }
}

extern "C" int lcv_unpatchify(const void* tok, float* out, int64_t B, int64_t Cout, int64_t T, int64_t H,
                              int64_t W, int tok_is_f32, void* stream) {
  LCV_CHECK_ARG(tok && out, "unpatchify: null pointer");
  LCV_CHECK_ARG(H % 2 == 0 && W % 2 == 0, "unpatchify: H, W must be even");
  const int64_t total = B * Cout * T * H * W;
  if (total == 0) return LCV_OK;
  int64_t blocks = (total + 255) / 256;
  if (blocks > 256 * 16) blocks = 256 * 16;
  if (tok_is_f32)
    hipLaunchKernelGGL(unpatchify_kernel<float>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream,
                       (const float*)tok, out, B, (int)Cout, (int)T, (int)H, (int)W);
  else
    hipLaunchKernelGGL(unpatchify_kernel<bf16_t>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream,
                       (const bf16_t*)tok, out, B, (int)Cout, (int)T, (int)H, (int)W);
  LCV_LAUNCH_CHECK("unpatchify");
  return LCV_OK;
}

// ---------------------------------------------------------------------------
// Denoise-step glue (fp32 latents)
// ---------------------------------------------------------------------------
#define LCV_CFG_PARTS 256   // workgroups (= partial sums) per sample of the zero-star dot products
// Partial <c, u> and <u, u> of one slice of a sample, summed in a FIXED order (lanes by the wave reduction, the four waves in
// index order): ws[b][part][2].  No atomics: two runs of a denoise step give the same bits (a hipGraph replay of the loop is
// compared bit for bit with the eager loop in tests/test_gpu_denoise_parity.py).
__global__ __launch_bounds__(256) void cfg_dot_kernel(const float* __restrict__ cond,
                                                      const float* __restrict__ uncond,
                                                      float* __restrict__ ws, int64_t n) {
  __shared__ float sd[4], sn[4];
  const int64_t b = blockIdx.y;
  const float* c = cond + b * n;
  const float* u = uncond + b * n;
  float dot = 0.f, nrm = 0.f;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const float cv = c[i], uv = u[i];
    dot += cv * uv;
    nrm += uv * uv;
  }
  dot = wave_sum(dot);
  nrm = wave_sum(nrm);
  if ((threadIdx.x & 63) == 0) { sd[threadIdx.x >> 6] = dot; sn[threadIdx.x >> 6] = nrm; }
  __syncthreads();
  if (threadIdx.x == 0) {
    float* o = ws + (b * LCV_CFG_PARTS + blockIdx.x) * 2;
    o[0] = ((sd[0] + sd[1]) + sd[2]) + sd[3];
    o[1] = ((sn[0] + sn[1]) + sn[2]) + sn[3];
  }
}

__global__ __launch_bounds__(256) void cfg_euler_kernel(const float* __restrict__ cond,
                                                        const float* __restrict__ uncond,
                                                        float* __restrict__ x, const float* __restrict__ ws,
                                                        int64_t n, float guidance, float dt, int negate,
                                                        int use_zero_star) {
  __shared__ float sd[4], sn[4];
  const int64_t b = blockIdx.y;
  float st = 1.0f;
  if (use_zero_star) {   // every workgroup adds the LCV_CFG_PARTS partial sums in the same order (2 KiB from L2)
    const float* o = ws + (b * LCV_CFG_PARTS + threadIdx.x) * 2;
    float d = wave_sum(o[0]), q = wave_sum(o[1]);
    if ((threadIdx.x & 63) == 0) { sd[threadIdx.x >> 6] = d; sn[threadIdx.x >> 6] = q; }
    __syncthreads();
    st = (((sd[0] + sd[1]) + sd[2]) + sd[3]) / ((((sn[0] + sn[1]) + sn[2]) + sn[3]) + 1e-8f);
  }
  const float sgn = negate ? -1.0f : 1.0f;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const float c = cond[b * n + i], u = uncond[b * n + i] * st;
    const float v = u + guidance * (c - u);
    x[b * n + i] += dt * (sgn * v);
  }
}

extern "C" int lcv_cfg_euler_step(const float* cond, const float* uncond, float* x, float* ws, int64_t B,
                                  int64_t n, float guidance, float dt, int negate, int use_zero_star,
                                  void* stream) {
  LCV_CHECK_ARG(cond && uncond && x && ws, "cfg_euler_step: null pointer");
  if (B == 0 || n == 0) return LCV_OK;
  hipStream_t s = (hipStream_t)stream;
  int64_t bx = (n + 255) / 256;
  if (bx > 1024) bx = 1024;
  if (use_zero_star) {   // always LCV_CFG_PARTS workgroups per sample: every partial-sum slot is written (empty slices write 0)
    hipLaunchKernelGGL(cfg_dot_kernel, dim3(LCV_CFG_PARTS, (unsigned)B), dim3(256), 0, s, cond, uncond, ws, n);
    LCV_LAUNCH_CHECK("cfg_dot");
  }
  hipLaunchKernelGGL(cfg_euler_kernel, dim3((unsigned)bx, (unsigned)B), dim3(256), 0, s, cond, uncond, x, ws,
                     n, guidance, dt, negate, use_zero_star);
  LCV_LAUNCH_CHECK("cfg_euler");
  return LCV_OK;
}

__global__ __launch_bounds__(256) void euler_kernel(const float* __restrict__ v, float* __restrict__ x,
                                                    int64_t n, float dt) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
    x[i] += dt * v[i];
}

extern "C" int lcv_euler_step(const float* v, float* x, int64_t n, float dt, int negate, void* stream) {
  LCV_CHECK_ARG(v && x, "euler_step: null pointer");
  if (n == 0) return LCV_OK;
  int64_t bx = (n + 255) / 256;
  if (bx > 2048) bx = 2048;
  hipLaunchKernelGGL(euler_kernel, dim3((unsigned)bx), dim3(256), 0, (hipStream_t)stream, v, x, n,
                     negate ? -dt : dt);
  LCV_LAUNCH_CHECK("euler_step");
  return LCV_OK;
}

// ---------------------------------------------------------------------------
// Flow-matching loss pieces (delta_experiment/scripts/common.py:458-488)
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void fm_noise_kernel(const bf16_t* __restrict__ x0,
                                                       const bf16_t* __restrict__ eps,
                                                       const float* __restrict__ sigma,
                                                       bf16_t* __restrict__ out, int64_t per_sample) {
  const int64_t b = blockIdx.y;
  const float s = sigma[b];
  const float oms = __fsub_rn(1.0f, s);
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < per_sample; i += (int64_t)gridDim.x * blockDim.x) {
    const float a = bf2f(x0[b * per_sample + i]);
    const float e = bf2f(eps[b * per_sample + i]);
    // (1-sigma)*x0 + sigma*eps with torch's three separately rounded fp32 ops, then .to(bf16)
    out[b * per_sample + i] = f2bf(__fadd_rn(__fmul_rn(oms, a), __fmul_rn(s, e)));
  }
}

extern "C" int lcv_fm_noise(const void* x0, const void* eps, const float* sigma, void* out, int64_t B,
                            int64_t per_sample, void* stream) {
  LCV_CHECK_ARG(x0 && eps && sigma && out, "fm_noise: null pointer");
  if (B == 0 || per_sample == 0) return LCV_OK;
  int64_t bx = (per_sample + 255) / 256;
  if (bx > 2048) bx = 2048;
  hipLaunchKernelGGL(fm_noise_kernel, dim3((unsigned)bx, (unsigned)B), dim3(256), 0, (hipStream_t)stream,
                     (const bf16_t*)x0, (const bf16_t*)eps, sigma, (bf16_t*)out, per_sample);
  LCV_LAUNCH_CHECK("fm_noise");
  return LCV_OK;
}

// Training loss.  Deterministic: every workgroup leaves ONE partial sum (wave sums added in wave order), a second one-workgroup
// launch adds the LCV_FM_MSE_BLOCKS partials in a fixed tree - no atomics, so two runs of the same step give the same float
// (the early stopper's strict `<` and the loss logs are reproducible).
__global__ __launch_bounds__(256) void fm_mse_kernel(const float* __restrict__ pred,
                                                     const bf16_t* __restrict__ eps,
                                                     const bf16_t* __restrict__ x0, float* __restrict__ parts,
                                                     float* __restrict__ dpred, int64_t BC, int T, int Tc,
                                                     int64_t HW, float inv_n) {
  const int Tt = T - Tc;
  const int64_t total = BC * T * HW;
  float acc = 0.f;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t hw = i % HW;
    const int64_t r = i / HW;
    const int t = (int)(r % T);
    const int64_t bc = r / T;
    float g = 0.f;
    if (t >= Tc) {
      const int64_t j = (bc * Tt + (t - Tc)) * HW + hw;
      // velocity target = (eps - x0) evaluated in bf16 then widened (common.py:486)
      const float vt = bfround(bf2f(eps[j]) - bf2f(x0[j]));
      const float d = pred[i] - vt;
      acc += d * d;
      g = 2.0f * d * inv_n;
    }
    if (dpred) dpred[i] = g;
  }
  __shared__ float red[4];
  acc = wave_sum(acc);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) parts[blockIdx.x] = ((red[0] + red[1]) + (red[2] + red[3])) * inv_n;
}

__global__ __launch_bounds__(LCV_FM_MSE_BLOCKS) void fm_mse_sum_kernel(const float* __restrict__ parts, float* __restrict__ loss,
                                                                       int n) {
  __shared__ float red[LCV_FM_MSE_BLOCKS];
  red[threadIdx.x] = (int)threadIdx.x < n ? parts[threadIdx.x] : 0.f;
  __syncthreads();
  for (int s = LCV_FM_MSE_BLOCKS / 2; s > 0; s >>= 1) {
    if ((int)threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
    __syncthreads();
  }
  if (threadIdx.x == 0) loss[0] = red[0];
}

extern "C" int lcv_fm_mse(const float* pred, const void* eps, const void* x0, float* loss_out, float* dpred, float* ws,
                          int64_t B, int64_t C, int64_t T, int64_t Tc, int64_t HW, void* stream) {
  LCV_CHECK_ARG(pred && eps && x0 && loss_out && ws, "fm_mse: null pointer");
  LCV_CHECK_ARG(T > Tc && Tc >= 0, "fm_mse: need at least one target frame (T=%ld, Tc=%ld)", (long)T, (long)Tc);
  hipStream_t s = (hipStream_t)stream;
  const int64_t total = B * C * T * HW;
  if (total == 0) {
    if (hipMemsetAsync(loss_out, 0, sizeof(float), s) != hipSuccess) {
      lcv_set_error("fm_mse: memset failed");
      return LCV_EDEVICE;
    }
    return LCV_OK;
  }
  const float inv_n = 1.0f / (float)(B * C * (T - Tc) * HW);
  int64_t bx = (total + 255) / 256;
  if (bx > LCV_FM_MSE_BLOCKS) bx = LCV_FM_MSE_BLOCKS;
  hipLaunchKernelGGL(fm_mse_kernel, dim3((unsigned)bx), dim3(256), 0, s, pred, (const bf16_t*)eps,
                     (const bf16_t*)x0, ws, dpred, B * C, (int)T, (int)Tc, HW, inv_n);
  LCV_LAUNCH_CHECK("fm_mse");
  hipLaunchKernelGGL(fm_mse_sum_kernel, dim3(1), dim3(LCV_FM_MSE_BLOCKS), 0, s, ws, loss_out, (int)bx);
  LCV_LAUNCH_CHECK("fm_mse (sum)");
  return LCV_OK;
}

// Per-sample form for the early stopper's anchor batch (sigmas x noise draws evaluated in ONE forward): loss[b] = mean over
// sample b's target slice, no gradient.  Deterministic: LCV_FM_MSE_PARTS fixed-order partial sums per sample, then one block
// per sample adds them in a fixed tree (no atomics), so a check gives the same floats on every run and the strict `<` of the
// stopper's bookkeeping is reproducible.  eps / x0 advance by `*_bstride` elements per sample (0 = shared by every sample).
__global__ __launch_bounds__(256) void fm_mse_parts_kernel(const float* __restrict__ pred, const bf16_t* __restrict__ eps,
                                                           const bf16_t* __restrict__ x0, float* __restrict__ parts,
                                                           int64_t C, int T, int Tc, int64_t HW, int64_t eps_bstride,
                                                           int64_t x0_bstride) {
  const int b = blockIdx.y;
  const int Tt = T - Tc;
  const int64_t per = C * Tt * HW;                       // target elements of one sample
  const float* pb = pred + (int64_t)b * C * T * HW;
  const bf16_t* eb = eps + (int64_t)b * eps_bstride;
  const bf16_t* xb = x0 + (int64_t)b * x0_bstride;
  float acc = 0.f;
  for (int64_t j = (int64_t)blockIdx.x * 256 + threadIdx.x; j < per; j += (int64_t)gridDim.x * 256) {
    const int64_t hw = j % HW;
    const int64_t r = j / HW;
    const int t = (int)(r % Tt);
    const int64_t c = r / Tt;
    const float vt = bfround(bf2f(eb[j]) - bf2f(xb[j]));
    const float d = pb[(c * T + Tc + t) * HW + hw] - vt;
    acc += d * d;
  }
  __shared__ float red[4];
  acc = wave_sum(acc);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) parts[(int64_t)b * gridDim.x + blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}

__global__ __launch_bounds__(LCV_FM_MSE_PARTS) void fm_mse_finish_kernel(const float* __restrict__ parts,
                                                                         float* __restrict__ loss, float inv_n) {
  __shared__ float red[LCV_FM_MSE_PARTS];
  red[threadIdx.x] = parts[(int64_t)blockIdx.x * LCV_FM_MSE_PARTS + threadIdx.x];
  __syncthreads();
  for (int s = LCV_FM_MSE_PARTS / 2; s > 0; s >>= 1) {
    if ((int)threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
    __syncthreads();
  }
  if (threadIdx.x == 0) loss[blockIdx.x] = red[0] * inv_n;
}

extern "C" int lcv_fm_mse_samples(const float* pred, const void* eps, const void* x0, float* loss_out, float* ws,
                                  int64_t B, int64_t C, int64_t T, int64_t Tc, int64_t HW, int64_t eps_bstride,
                                  int64_t x0_bstride, void* stream) {
  LCV_CHECK_ARG(pred && eps && x0 && loss_out && ws, "fm_mse_samples: null pointer");
  LCV_CHECK_ARG(T > Tc && Tc >= 0, "fm_mse_samples: need at least one target frame (T=%ld, Tc=%ld)", (long)T, (long)Tc);
  LCV_CHECK_ARG(B >= 0 && B <= 65535, "fm_mse_samples: batch %ld out of range", (long)B);
  const int64_t per = C * (T - Tc) * HW;
  LCV_CHECK_ARG((eps_bstride == 0 || eps_bstride >= per) && (x0_bstride == 0 || x0_bstride >= per),
                "fm_mse_samples: sample strides overlap (per-sample target elements %ld)", (long)per);
  if (B == 0 || per == 0) return LCV_OK;
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(fm_mse_parts_kernel, dim3(LCV_FM_MSE_PARTS, (unsigned)B), dim3(256), 0, s, pred, (const bf16_t*)eps,
                     (const bf16_t*)x0, ws, C, (int)T, (int)Tc, HW, eps_bstride, x0_bstride);
  LCV_LAUNCH_CHECK("fm_mse_samples (parts)");
  hipLaunchKernelGGL(fm_mse_finish_kernel, dim3((unsigned)B), dim3(LCV_FM_MSE_PARTS), 0, s, ws, loss_out,
                     1.0f / (float)per);
  LCV_LAUNCH_CHECK("fm_mse_samples (finish)");
  return LCV_OK;
}

// ---------------------------------------------------------------------------
// Sinusoidal timestep features (the input of the timestep MLP): out[i, j] = cos(t_i f_j) for j < half, sin(t_i f_j) after,
// f_j = exp(-ln(max_period) j / half), all fp32 — the one piece of arithmetic the DiT forward still did in torch.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void timestep_embedding_kernel(const float* __restrict__ t, float* __restrict__ out, int n,
                                                                 int half, float neg_log_period_over_half) {
  const int idx = blockIdx.x * 256 + threadIdx.x;
  if (idx >= n * half) return;
  const int i = idx / half, j = idx - i * half;
  const float arg = t[i] * expf(neg_log_period_over_half * (float)j);
  out[(int64_t)i * 2 * half + j] = cosf(arg);
  out[(int64_t)i * 2 * half + half + j] = sinf(arg);
}

extern "C" int lcv_timestep_embedding(const float* t, float* out, int64_t n, int64_t dim, float max_period, void* stream) {
  LCV_CHECK_ARG(t && out && n >= 0 && dim > 0 && dim % 2 == 0 && max_period > 1.f, "timestep_embedding: bad arguments");
  if (n == 0) return LCV_OK;
  const int half = (int)(dim / 2);
  hipLaunchKernelGGL(timestep_embedding_kernel, dim3((unsigned)((n * half + 255) / 256)), dim3(256), 0, (hipStream_t)stream, t, out,
                     (int)n, half, -logf(max_period) / (float)half);
  LCV_LAUNCH_CHECK("timestep_embedding");
  return LCV_OK;
}
