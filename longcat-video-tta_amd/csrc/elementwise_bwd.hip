// Backward kernels of the HBM-bound block pieces (TTA inner loop): AdaLN / LayerNorm, gated residual,
// q/k RMSNorm + RoPE, SwiGLU, unpatchify, the fp32 small-M linear, and the skinny token contraction
// that yields the LoRA dA / dB.  bf16 roundings of the forward are treated as identity (straight-through),
// as autograd does for the reference's bf16 modules.
#include "lcv_common.h"

#define ROWNORM_MAXCH 8

// ---------------------------------------------------------------------------
// LayerNorm backward, one wave per row.
//   MODE 0 (AdaLN): y = xh*(1+scale)+shift ; g = dy*(1+scale) ; dshift += dy ; dscale += dy*xh
//   MODE 1 (affine): y = xh*w+b            ; g = dy*w         ; db += dy     ; dw += dy*xh
//   dx = rstd * (g - mean(g) - xh * mean(g*xh))
// ---------------------------------------------------------------------------
// Rows per workgroup when the parameter / modulation gradients are wanted: each wave walks ROWNORM_RPB/4 rows and the
// per-channel sums of the whole workgroup are kept in LDS (ds_add_f32), so the global fp32 atomics are one per channel per
// ROWNORM_RPB rows instead of one per channel per row (51 M -> 1.6 M per call at 6 240 tokens: the kernel was atomic-bound).
#define ROWNORM_RPB 32

template <int MODE>
__global__ __launch_bounds__(256) void rownorm_bwd_kernel(
    const bf16_t* __restrict__ x, const float* __restrict__ p_mul, const bf16_t* __restrict__ dy,
    bf16_t* __restrict__ dx, float* __restrict__ d_add, float* __restrict__ d_mul, int64_t rows, int C,
    int64_t S, int64_t mod_stride, float eps, int rows_per_block, const bf16_t* __restrict__ dres) {
  __shared__ float s_add[ROWNORM_MAXCH * 64 * 8], s_mul[ROWNORM_MAXCH * 64 * 8];
  const int lane = threadIdx.x & 63;
  const bool want_d = d_add != nullptr;
  const int64_t row0 = (int64_t)blockIdx.x * rows_per_block;
  const int64_t frame0 = (MODE == 0) ? row0 / S : 0;          // the frame the LDS sums belong to
  if (want_d) {
    for (int i = threadIdx.x; i < C; i += 256) { s_add[i] = 0.f; s_mul[i] = 0.f; }
    __syncthreads();
  }
  for (int it = 0; it < rows_per_block; it += 4) {
    const int64_t row = row0 + it + (threadIdx.x >> 6);
    if (row >= rows) continue;
  const bf16_t* xr = x + row * C;
  const bf16_t* gr = dy + row * C;
  const int64_t frame = (MODE == 0) ? row / S : 0;
  const float* pm = p_mul + frame * mod_stride;
  float v[ROWNORM_MAXCH][8], g[ROWNORM_MAXCH][8];
  float sum = 0.f;
#pragma unroll
  for (int ch = 0; ch < ROWNORM_MAXCH; ++ch) {
    const int c = (ch * 64 + lane) * 8;
    if (c < C) {
      unpack8(*reinterpret_cast<const u16x8*>(xr + c), v[ch]);
      unpack8(*reinterpret_cast<const u16x8*>(gr + c), g[ch]);
#pragma unroll
      for (int i = 0; i < 8; ++i) sum += v[ch][i];
    } else {
#pragma unroll
      for (int i = 0; i < 8; ++i) { v[ch][i] = 0.f; g[ch][i] = 0.f; }
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
  float sg = 0.f, sgx = 0.f;
  const bool in_lds = frame == frame0;      // a workgroup that straddles two frames sends the second one's rows directly
#pragma unroll
  for (int ch = 0; ch < ROWNORM_MAXCH; ++ch) {
    const int c = (ch * 64 + lane) * 8;
    if (c < C) {
      const f32x4 m0 = *reinterpret_cast<const f32x4*>(pm + c);
      const f32x4 m1 = *reinterpret_cast<const f32x4*>(pm + c + 4);
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const float xh = (v[ch][i] - mean) * rstd;
        const float dyv = g[ch][i];
        if (want_d) {  // parameter / modulation gradients
          if (in_lds) {
            atomicAdd(&s_add[c + i], dyv);
            atomicAdd(&s_mul[c + i], dyv * xh);
          } else {
            atomicAdd(d_add + frame * mod_stride + c + i, dyv);
            atomicAdd(d_mul + frame * mod_stride + c + i, dyv * xh);
          }
        }
        const float mul = ((i < 4) ? m0[i] : m1[i - 4]) + ((MODE == 0) ? 1.0f : 0.0f);
        const float gg = dyv * mul;
        v[ch][i] = xh;
        g[ch][i] = gg;
        sg += gg;
        sgx += gg * xh;
      }
    }
  }
  const float mg = wave_sum(sg) / (float)C;
  const float mgx = wave_sum(sgx) / (float)C;
  bf16_t* dr = dx + row * C;
#pragma unroll
  for (int ch = 0; ch < ROWNORM_MAXCH; ++ch) {
    const int c = (ch * 64 + lane) * 8;
    if (c < C) {
      float o[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) o[i] = rstd * (g[ch][i] - mg - v[ch][i] * mgx);
      if (dres) {   // the gradient that reaches x through the residual path of the same block, summed here in fp32
        float rr[8];
        unpack8(*reinterpret_cast<const u16x8*>(dres + row * C + c), rr);
#pragma unroll
        for (int i = 0; i < 8; ++i) o[i] += rr[i];
      }
      *reinterpret_cast<u16x8*>(dr + c) = pack8(o);
    }
  }
  }
  if (want_d) {
    __syncthreads();
    for (int i = threadIdx.x; i < C; i += 256) {
      atomicAdd(d_add + frame0 * mod_stride + i, s_add[i]);
      atomicAdd(d_mul + frame0 * mod_stride + i, s_mul[i]);
    }
  }
}

extern "C" int lcv_adaln_modulate_bwd(const void* x, const float* mod, const void* dy, void* dx, float* dmod,
                                      int64_t B, int64_t T, int64_t S, int64_t C, int64_t mod_stride,
                                      int64_t shift_off, int64_t scale_off, float eps, const void* dres, void* stream) {
  LCV_CHECK_ARG(x && mod && dy && dx, "adaln_modulate_bwd: null pointer");
  LCV_CHECK_ARG(C > 0 && C % 8 == 0 && C <= 4096, "adaln_modulate_bwd: C must be a multiple of 8 and <= 4096");
  const int64_t rows = B * T * S;
  if (rows == 0) return LCV_OK;
  const int rpb = dmod ? ROWNORM_RPB : 4;
  hipLaunchKernelGGL(rownorm_bwd_kernel<0>, dim3((unsigned)((rows + rpb - 1) / rpb)), dim3(256), 0, (hipStream_t)stream,
                     (const bf16_t*)x, mod + scale_off, (const bf16_t*)dy, (bf16_t*)dx,
                     dmod ? dmod + shift_off : nullptr, dmod ? dmod + scale_off : nullptr, rows, (int)C, S,
                     mod_stride, eps, rpb, (const bf16_t*)dres);
  LCV_LAUNCH_CHECK("adaln_modulate_bwd");
  return LCV_OK;
}

extern "C" int lcv_layernorm_affine_bwd(const void* x, const float* w, const void* dy, void* dx, float* dw,
                                        float* db, int64_t rows, int64_t C, float eps, const void* dres, void* stream) {
  LCV_CHECK_ARG(x && w && dy && dx, "layernorm_affine_bwd: null pointer");
  LCV_CHECK_ARG((dw == nullptr) == (db == nullptr), "layernorm_affine_bwd: dw and db go together");
  LCV_CHECK_ARG(C > 0 && C % 8 == 0 && C <= 4096, "layernorm_affine_bwd: C must be a multiple of 8 and <= 4096");
  if (rows == 0) return LCV_OK;
  const int rpb = dw ? ROWNORM_RPB : 4;
  hipLaunchKernelGGL(rownorm_bwd_kernel<1>, dim3((unsigned)((rows + rpb - 1) / rpb)), dim3(256), 0, (hipStream_t)stream,
                     (const bf16_t*)x, w, (const bf16_t*)dy, (bf16_t*)dx, db, dw, rows, (int)C, (int64_t)1,
                     (int64_t)0, eps, rpb, (const bf16_t*)dres);
  LCV_LAUNCH_CHECK("layernorm_affine_bwd");
  return LCV_OK;
}

// ---------------------------------------------------------------------------
// gated residual backward: dy = gate * dout ; dgate[frame] += dout * y
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void gate_residual_bwd_kernel(const bf16_t* __restrict__ y,
                                                                const float* __restrict__ gate,
                                                                const bf16_t* __restrict__ dout,
                                                                bf16_t* __restrict__ dy, float* __restrict__ dgate,
                                                                int64_t n_packets, int cpk, int64_t S,
                                                                int64_t mod_stride) {
  for (int64_t pk = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; pk < n_packets;
       pk += (int64_t)gridDim.x * blockDim.x) {
    const int64_t row = pk / cpk;
    const int c = (int)(pk - row * cpk) * 8;
    const int64_t goff = (row / S) * mod_stride + c;
    float d[8], o[8];
    unpack8(*reinterpret_cast<const u16x8*>(dout + pk * 8), d);
    const f32x4 g0 = *reinterpret_cast<const f32x4*>(gate + goff);
    const f32x4 g1 = *reinterpret_cast<const f32x4*>(gate + goff + 4);
#pragma unroll
    for (int i = 0; i < 8; ++i) o[i] = d[i] * ((i < 4) ? g0[i] : g1[i - 4]);
    *reinterpret_cast<u16x8*>(dy + pk * 8) = pack8(o);
    if (dgate) {
      float yf[8];
      unpack8(*reinterpret_cast<const u16x8*>(y + pk * 8), yf);
#pragma unroll
      for (int i = 0; i < 8; ++i) atomicAdd(dgate + goff + i, d[i] * yf[i]);
    }
  }
}

// dgate wanted: a workgroup owns GATE_RPB consecutive rows; thread t owns the same 8-channel packets (t, t+256, ...) in
// every row, so the per-channel products accumulate in registers and reach memory as one atomic per channel per
// GATE_RPB rows (flushed early when the rows cross into the next frame).
#define GATE_RPB 32
#define GATE_MAXPK 2   // C <= 4096: 512 packets per row over 256 threads
__global__ __launch_bounds__(256) void gate_residual_bwd_dgate_kernel(const bf16_t* __restrict__ y, const float* __restrict__ gate,
                                                                      const bf16_t* __restrict__ dout, bf16_t* __restrict__ dy,
                                                                      float* __restrict__ dgate, int64_t rows, int cpk, int64_t S,
                                                                      int64_t mod_stride) {
  const int64_t row0 = (int64_t)blockIdx.x * GATE_RPB;
  float acc[GATE_MAXPK][8];
#pragma unroll
  for (int u = 0; u < GATE_MAXPK; ++u)
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[u][i] = 0.f;
  int64_t cur_frame = row0 / S;
  auto flush = [&](int64_t frame) {
#pragma unroll
    for (int u = 0; u < GATE_MAXPK; ++u) {
      const int pkc = threadIdx.x + u * 256;
      if (pkc < cpk) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          atomicAdd(dgate + frame * mod_stride + pkc * 8 + i, acc[u][i]);
          acc[u][i] = 0.f;
        }
      }
    }
  };
  for (int r = 0; r < GATE_RPB; ++r) {
    const int64_t row = row0 + r;
    if (row >= rows) break;
    const int64_t frame = row / S;
    if (frame != cur_frame) { flush(cur_frame); cur_frame = frame; }
#pragma unroll
    for (int u = 0; u < GATE_MAXPK; ++u) {
      const int pkc = threadIdx.x + u * 256;
      if (pkc < cpk) {
        const int64_t pk = row * cpk + pkc;
        const int64_t goff = frame * mod_stride + pkc * 8;
        float d[8], o[8], yf[8];
        unpack8(*reinterpret_cast<const u16x8*>(dout + pk * 8), d);
        unpack8(*reinterpret_cast<const u16x8*>(y + pk * 8), yf);
        const f32x4 g0 = *reinterpret_cast<const f32x4*>(gate + goff);
        const f32x4 g1 = *reinterpret_cast<const f32x4*>(gate + goff + 4);
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          o[i] = d[i] * ((i < 4) ? g0[i] : g1[i - 4]);
          acc[u][i] = fmaf(d[i], yf[i], acc[u][i]);
        }
        *reinterpret_cast<u16x8*>(dy + pk * 8) = pack8(o);
      }
    }
  }
  flush(cur_frame);
}

extern "C" int lcv_gate_residual_bwd(const void* y, const float* mod, const void* dout, void* dy, float* dmod,
                                     int64_t B, int64_t T, int64_t S, int64_t C, int64_t mod_stride,
                                     int64_t gate_off, void* stream) {
  LCV_CHECK_ARG(y && mod && dout && dy, "gate_residual_bwd: null pointer");
  LCV_CHECK_ARG(C % 8 == 0 && gate_off % 4 == 0 && mod_stride % 4 == 0, "gate_residual_bwd: bad alignment");
  const int64_t n_packets = B * T * S * (C / 8);
  if (n_packets == 0) return LCV_OK;
  if (dmod && C / 8 <= 256 * GATE_MAXPK) {
    const int64_t rows = B * T * S;
    hipLaunchKernelGGL(gate_residual_bwd_dgate_kernel, dim3((unsigned)((rows + GATE_RPB - 1) / GATE_RPB)), dim3(256), 0,
                       (hipStream_t)stream, (const bf16_t*)y, mod + gate_off, (const bf16_t*)dout, (bf16_t*)dy, dmod + gate_off,
                       rows, (int)(C / 8), S, mod_stride);
    LCV_LAUNCH_CHECK("gate_residual_bwd");
    return LCV_OK;
  }
  int64_t blocks = (n_packets + 255) / 256;
  if (blocks > 256 * 16) blocks = 256 * 16;
  hipLaunchKernelGGL(gate_residual_bwd_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream,
                     (const bf16_t*)y, mod + gate_off, (const bf16_t*)dout, (bf16_t*)dy,
                     dmod ? dmod + gate_off : nullptr, n_packets, (int)(C / 8), S, mod_stride);
  LCV_LAUNCH_CHECK("gate_residual_bwd");
  return LCV_OK;
}

// ---------------------------------------------------------------------------
// q/k RMSNorm + RoPE backward (weights frozen):  dx = r * (dn - n * mean(dn * n)),  dn = w * rope^T(dout)
// ---------------------------------------------------------------------------
__device__ __forceinline__ void norm_rope_bwd_vec(const bf16_t* xin, const bf16_t* dout, bf16_t* dxin,
                                                  const float (&w)[8], const float (&cs)[8], bool do_rope,
                                                  float eps, float out_scale, float (&dwacc)[8], bool want_dw) {
  float x[8], d[8];
  unpack8(*reinterpret_cast<const u16x8*>(xin), x);
  unpack8(*reinterpret_cast<const u16x8*>(dout), d);
#pragma unroll
  for (int i = 0; i < 8; ++i) d[i] *= out_scale;  // the forward multiplied its output by out_scale
  float ss = 0.f;
#pragma unroll
  for (int i = 0; i < 8; ++i) ss += x[i] * x[i];
  ss += __shfl_xor(ss, 8, 64);
  ss += __shfl_xor(ss, 4, 64);
  ss += __shfl_xor(ss, 2, 64);
  ss += __shfl_xor(ss, 1, 64);
  const float r = rsqrtf(ss * (1.0f / 128.0f) + eps);
  float dn[8], n[8];
  float dot = 0.f;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    float d0 = d[2 * i], d1 = d[2 * i + 1];
    if (do_rope) {
      const float c = cs[2 * i], s = cs[2 * i + 1];
      const float t0 = d0 * c + d1 * s;
      const float t1 = d1 * c - d0 * s;
      d0 = t0;
      d1 = t1;
    }
    dn[2 * i] = d0 * w[2 * i];
    dn[2 * i + 1] = d1 * w[2 * i + 1];
    if (want_dw) {  // y = rope(n * w): dw += rope^T(dout) * n (norm-weight tuning, run_norm_tune_tta.py:87-98)
      dwacc[2 * i] += d0 * (x[2 * i] * r);
      dwacc[2 * i + 1] += d1 * (x[2 * i + 1] * r);
    }
  }
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    n[i] = x[i] * r;
    dot += dn[i] * n[i];
  }
  dot += __shfl_xor(dot, 8, 64);
  dot += __shfl_xor(dot, 4, 64);
  dot += __shfl_xor(dot, 2, 64);
  dot += __shfl_xor(dot, 1, 64);
  dot *= (1.0f / 128.0f);
  float o[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) o[i] = r * (dn[i] - n[i] * dot);
  *reinterpret_cast<u16x8*>(dxin) = pack8(o);
}

__global__ __launch_bounds__(256) void qknorm_rope_bwd_kernel(
    const bf16_t* __restrict__ q_in, const bf16_t* __restrict__ k_in, const bf16_t* __restrict__ dq_out,
    const bf16_t* __restrict__ dk_out, bf16_t* __restrict__ dq_in, bf16_t* __restrict__ dk_in,
    const bf16_t* __restrict__ wq, const bf16_t* __restrict__ wk, const float* __restrict__ cs_tab, int H,
    int64_t in_sb, int64_t in_sn, int64_t q_sb, int64_t q_sn, int64_t kv_sb, int64_t kv_sn, int64_t din_sb,
    int64_t din_sn, int64_t pos_off, float eps, float q_scale, float* __restrict__ dwq, float* __restrict__ dwk,
    int dw_slots) {
  __shared__ float s_dw[2][4][128];
  const int64_t n = blockIdx.x, b = blockIdx.y;
  const int sub = threadIdx.x & 15;
  const int hl = threadIdx.x >> 4;
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
  float dwq_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, dwk_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  for (int h0 = 0; h0 < H; h0 += 16) {
    const int h = h0 + hl;
    if (h >= H) continue;
    const int64_t off = (int64_t)h * 128 + sub * 8;
    if (q_in)
      norm_rope_bwd_vec(q_in + b * in_sb + n * in_sn + off, dq_out + b * q_sb + n * q_sn + off,
                        dq_in + b * din_sb + n * din_sn + off, wqf, cs, do_rope, eps, q_scale, dwq_acc, dwq != nullptr);
    if (k_in)
      norm_rope_bwd_vec(k_in + b * in_sb + n * in_sn + off, dk_out + b * kv_sb + n * kv_sn + off,
                        dk_in + b * din_sb + n * din_sn + off, wkf, cs, do_rope, eps, 1.0f, dwk_acc, dwk != nullptr);
  }
  // norm-weight gradients (only when asked for): sum over the wave's 4 heads-in-flight, then over the 4 waves through
  // LDS, then ONE atomic per dim per workgroup into accumulator row (token % dw_slots) — every token of every layer
  // call lands on the same 128 floats otherwise, and the kernel becomes a queue on those addresses
  if (dwq != nullptr || dwk != nullptr) {
    const int wave = threadIdx.x >> 6;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      float a = dwq_acc[i], c = dwk_acc[i];
      a += __shfl_xor(a, 16, 64); a += __shfl_xor(a, 32, 64);
      c += __shfl_xor(c, 16, 64); c += __shfl_xor(c, 32, 64);
      if ((threadIdx.x & 63) < 16) {
        s_dw[0][wave][sub * 8 + i] = a;
        s_dw[1][wave][sub * 8 + i] = c;
      }
    }
    __syncthreads();
    const int slot = (int)((n + b * gridDim.x) % dw_slots);
    if (threadIdx.x < 128) {
      const int d = threadIdx.x;
      if (dwq != nullptr && q_in) atomicAdd(dwq + slot * 128 + d, s_dw[0][0][d] + s_dw[0][1][d] + s_dw[0][2][d] + s_dw[0][3][d]);
    } else {
      const int d = threadIdx.x - 128;
      if (dwk != nullptr && k_in) atomicAdd(dwk + slot * 128 + d, s_dw[1][0][d] + s_dw[1][1][d] + s_dw[1][2][d] + s_dw[1][3][d]);
    }
  }
}

extern "C" int lcv_qknorm_rope_bwd(const void* q_in, const void* k_in, const void* dq_out, const void* dk_out,
                                   void* dq_in, void* dk_in, const void* wq, const void* wk, const void* cs,
                                   int64_t B, int64_t N, int64_t H, int64_t in_sb, int64_t in_sn, int64_t q_sb,
                                   int64_t q_sn, int64_t kv_sb, int64_t kv_sn, int64_t din_sb, int64_t din_sn,
                                   int64_t pos_off, float eps, float q_scale, float* dwq, float* dwk, int64_t dw_slots,
                                   void* stream) {
  LCV_CHECK_ARG((q_in || k_in) && wq && wk, "qknorm_rope_bwd: null pointer");
  LCV_CHECK_ARG(!q_in || (dq_out && dq_in), "qknorm_rope_bwd: q gradients missing");
  LCV_CHECK_ARG(!k_in || (dk_out && dk_in), "qknorm_rope_bwd: k gradients missing");
  LCV_CHECK_ARG(in_sn % 8 == 0 && q_sn % 8 == 0 && kv_sn % 8 == 0 && din_sn % 8 == 0, "qknorm_rope_bwd: strides % 8");
  LCV_CHECK_ARG((!dwq && !dwk) || (dw_slots >= 1 && dw_slots <= 65536), "qknorm_rope_bwd: dw_slots");
  if (B == 0 || N == 0) return LCV_OK;
  hipLaunchKernelGGL(qknorm_rope_bwd_kernel, dim3((unsigned)N, (unsigned)B), dim3(256), 0, (hipStream_t)stream,
                     (const bf16_t*)q_in, (const bf16_t*)k_in, (const bf16_t*)dq_out, (const bf16_t*)dk_out,
                     (bf16_t*)dq_in, (bf16_t*)dk_in, (const bf16_t*)wq, (const bf16_t*)wk, (const float*)cs,
                     (int)H, in_sb, in_sn, q_sb, q_sn, kv_sb, kv_sn, din_sb, din_sn, pos_off, eps, q_scale, dwq, dwk,
                     (int)(dw_slots < 1 ? 1 : dw_slots));
  LCV_LAUNCH_CHECK("qknorm_rope_bwd");
  return LCV_OK;
}

// ---------------------------------------------------------------------------
// SwiGLU backward
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void swiglu_bwd_kernel(const bf16_t* __restrict__ gate,
                                                         const bf16_t* __restrict__ up,
                                                         const bf16_t* __restrict__ dout,
                                                         bf16_t* __restrict__ dgate, bf16_t* __restrict__ dup,
                                                         int64_t rows, int fpk, int64_t ld_in) {
  const int64_t n_packets = rows * fpk;
  for (int64_t pk = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; pk < n_packets;
       pk += (int64_t)gridDim.x * blockDim.x) {
    const int64_t row = pk / fpk;
    const int c = (int)(pk - row * fpk) * 8;
    float g[8], u[8], d[8], og[8], ou[8];
    unpack8(*reinterpret_cast<const u16x8*>(gate + row * ld_in + c), g);
    unpack8(*reinterpret_cast<const u16x8*>(up + row * ld_in + c), u);
    unpack8(*reinterpret_cast<const u16x8*>(dout + pk * 8), d);
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const float sig = 1.0f / (1.0f + __expf(-g[i]));
      const float s = g[i] * sig;
      ou[i] = d[i] * bfround(s);
      og[i] = d[i] * u[i] * (sig * (1.0f + g[i] * (1.0f - sig)));
    }
    *reinterpret_cast<u16x8*>(dgate + pk * 8) = pack8(og);
    *reinterpret_cast<u16x8*>(dup + pk * 8) = pack8(ou);
  }
}

// The same on the fused GEMM's layout: `gu` [rows, 2F] holds, per 64 columns, 32 gate values then their 32 up partners
// (LCV_EPI_SWIGLU's auxiliary output); `dgu` gets the gradients in that layout, ready to be the A operand of ONE GEMM
// against the transposed interleaved weight (dx = dgate W1 + dup W3 without a second GEMM and an add).
__global__ __launch_bounds__(256) void swiglu_bwd_il_kernel(const bf16_t* __restrict__ gu, const bf16_t* __restrict__ dout,
                                                            bf16_t* __restrict__ dgu, int64_t rows, int fpk) {
  const int64_t n_packets = rows * fpk;
  for (int64_t pk = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; pk < n_packets;
       pk += (int64_t)gridDim.x * blockDim.x) {
    const int64_t row = pk / fpk;
    const int c = (int)(pk - row * fpk) * 8;                       // feature index
    const int64_t at = row * (int64_t)(fpk * 16) + (c >> 5) * 64 + (c & 31);   // its gate column in the interleaved row
    float g[8], u[8], d[8], og[8], ou[8];
    unpack8(*reinterpret_cast<const u16x8*>(gu + at), g);
    unpack8(*reinterpret_cast<const u16x8*>(gu + at + 32), u);
    unpack8(*reinterpret_cast<const u16x8*>(dout + pk * 8), d);
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const float sig = 1.0f / (1.0f + __expf(-g[i]));
      const float s = g[i] * sig;
      ou[i] = d[i] * bfround(s);
      og[i] = d[i] * u[i] * (sig * (1.0f + g[i] * (1.0f - sig)));
    }
    *reinterpret_cast<u16x8*>(dgu + at) = pack8(og);
    *reinterpret_cast<u16x8*>(dgu + at + 32) = pack8(ou);
  }
}

extern "C" int lcv_swiglu_bwd_interleaved(const void* gu, const void* dout, void* dgu, int64_t rows, int64_t F, void* stream) {
  LCV_CHECK_ARG(gu && dout && dgu, "swiglu_bwd_interleaved: null pointer");
  LCV_CHECK_ARG(F % 32 == 0, "swiglu_bwd_interleaved: F=%ld must be a multiple of 32", (long)F);
  const int64_t n_packets = rows * (F / 8);
  if (n_packets == 0) return LCV_OK;
  int64_t blocks = (n_packets + 255) / 256;
  if (blocks > 256 * 16) blocks = 256 * 16;
  hipLaunchKernelGGL(swiglu_bwd_il_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)gu,
                     (const bf16_t*)dout, (bf16_t*)dgu, rows, (int)(F / 8));
  LCV_LAUNCH_CHECK("swiglu_bwd_interleaved");
  return LCV_OK;
}

extern "C" int lcv_swiglu_bwd(const void* gate, const void* up, const void* dout, void* dgate, void* dup,
                              int64_t rows, int64_t F, int64_t ld_in, void* stream) {
  LCV_CHECK_ARG(gate && up && dout && dgate && dup, "swiglu_bwd: null pointer");
  LCV_CHECK_ARG(F % 8 == 0 && ld_in % 8 == 0, "swiglu_bwd: F and ld_in must be multiples of 8");
  const int64_t n_packets = rows * (F / 8);
  if (n_packets == 0) return LCV_OK;
  int64_t blocks = (n_packets + 255) / 256;
  if (blocks > 256 * 16) blocks = 256 * 16;
  hipLaunchKernelGGL(swiglu_bwd_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream,
                     (const bf16_t*)gate, (const bf16_t*)up, (const bf16_t*)dout, (bf16_t*)dgate, (bf16_t*)dup, rows,
                     (int)(F / 8), ld_in);
  LCV_LAUNCH_CHECK("swiglu_bwd");
  return LCV_OK;
}

// ---------------------------------------------------------------------------
// unpatchify backward: dout [B,Cout,T,H,W] fp32 -> dtok [B, N, (ph pw c)] fp32
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void unpatchify_bwd_kernel(const float* __restrict__ dout,
                                                             float* __restrict__ dtok, int64_t B, int Cout, int T,
                                                             int H, int W) {
  const int Hh = H / 2, Wh = W / 2;
  const int64_t total = (int64_t)B * T * Hh * Wh * 4 * Cout;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (int64_t)gridDim.x * blockDim.x) {
    const int k = (int)(i % (4 * Cout));
    int64_t tkn = i / (4 * Cout);
    const int c = k % Cout;
    const int pp = k / Cout;  // ph*2 + pw
    const int wq = (int)(tkn % Wh);
    int64_t r = tkn / Wh;
    const int hq = (int)(r % Hh);
    r /= Hh;
    const int t = (int)(r % T);
    const int64_t b = r / T;
    dtok[i] = dout[(((b * Cout + c) * T + t) * H + 2 * hq + (pp >> 1)) * (int64_t)W + 2 * wq + (pp & 1)];
  }
}

extern "C" int lcv_unpatchify_bwd(const float* dout, float* dtok, int64_t B, int64_t Cout, int64_t T, int64_t H,
                                  int64_t W, void* stream) {
  LCV_CHECK_ARG(dout && dtok, "unpatchify_bwd: null pointer");
  const int64_t total = B * Cout * T * H * W;
  if (total == 0) return LCV_OK;
  int64_t blocks = (total + 255) / 256;
  if (blocks > 256 * 16) blocks = 256 * 16;
  hipLaunchKernelGGL(unpatchify_bwd_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, dout, dtok,
                     B, (int)Cout, (int)T, (int)H, (int)W);
  LCV_LAUNCH_CHECK("unpatchify_bwd");
  return LCV_OK;
}

// ---------------------------------------------------------------------------
// fp32 small-M linear backward w.r.t. its input:  da[m,k] = act'(a[m,k]) * sum_n dy[m,n] W[n,k]
// Partial sums over 256-row slabs of W are accumulated with fp32 atomics into `da` (zero-filled here);
// a second tiny launch applies the SiLU derivative.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void linear_f32_smallm_bwd_kernel(const float* __restrict__ dy,
                                                                    const bf16_t* __restrict__ w,
                                                                    float* __restrict__ da, int M, int64_t N, int K,
                                                                    int nchunk) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  float* sdy = reinterpret_cast<float*>(smem);  // [M][nchunk]
  const int64_t n0 = (int64_t)blockIdx.x * nchunk;
  const int nn = (int)((N - n0) < nchunk ? (N - n0) : nchunk);
  for (int i = threadIdx.x; i < M * nchunk; i += 256) {
    const int m = i / nchunk, j = i - m * nchunk;
    sdy[i] = (j < nn) ? dy[(int64_t)m * N + n0 + j] : 0.f;
  }
  __syncthreads();
  for (int k0 = threadIdx.x * 2; k0 < K; k0 += 512) {
    float acc0[16], acc1[16];
#pragma unroll
    for (int m = 0; m < 16; ++m) { acc0[m] = 0.f; acc1[m] = 0.f; }
    for (int j = 0; j < nn; ++j) {
      const u16x2 wv = *reinterpret_cast<const u16x2*>(w + (n0 + j) * K + k0);
      const float w0 = bf2f(wv[0]), w1 = bf2f(wv[1]);
#pragma unroll
      for (int m = 0; m < 16; ++m) {
        if (m < M) {
          const float d = sdy[m * nchunk + j];
          acc0[m] += d * w0;
          acc1[m] += d * w1;
        }
      }
    }
#pragma unroll
    for (int m = 0; m < 16; ++m) {
      if (m < M) {
        atomicAdd(da + (int64_t)m * K + k0, acc0[m]);
        atomicAdd(da + (int64_t)m * K + k0 + 1, acc1[m]);
      }
    }
  }
}

__global__ __launch_bounds__(256) void silu_grad_kernel(const float* __restrict__ a, float* __restrict__ da,
                                                        int64_t n) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) {
    const float x = a[i];
    const float sig = 1.0f / (1.0f + __expf(-x));
    da[i] *= sig * (1.0f + x * (1.0f - sig));
  }
}

extern "C" int lcv_linear_f32_smallm_bwd(const float* dy, const void* w, const float* a, float* da, int64_t M,
                                         int64_t N, int64_t K, int act_in, void* stream) {
  LCV_CHECK_ARG(dy && w && a && da, "linear_f32_smallm_bwd: null pointer");
  LCV_CHECK_ARG(K % 2 == 0, "linear_f32_smallm_bwd: K must be even");
  hipStream_t s = (hipStream_t)stream;
  if (hipMemsetAsync(da, 0, sizeof(float) * M * K, s) != hipSuccess) {
    lcv_set_error("linear_f32_smallm_bwd: memset failed");
    return LCV_EDEVICE;
  }
  const int nchunk = 256;
  for (int64_t m0 = 0; m0 < M; m0 += 16) {
    const int Mc = (int)((M - m0) < 16 ? (M - m0) : 16);
    hipLaunchKernelGGL(linear_f32_smallm_bwd_kernel, dim3((unsigned)((N + nchunk - 1) / nchunk)), dim3(256),
                       (size_t)Mc * nchunk * 4, s, dy + m0 * N, (const bf16_t*)w, da + m0 * K, Mc, N, (int)K, nchunk);
    LCV_LAUNCH_CHECK("linear_f32_smallm_bwd");
  }
  if (act_in == 1) {
    const int64_t n = M * K;
    hipLaunchKernelGGL(silu_grad_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, a, da, n);
    LCV_LAUNCH_CHECK("silu_grad");
  }
  return LCV_OK;
}

// ---------------------------------------------------------------------------
// Skinny token contraction for the LoRA parameter gradients:
//   out[r, k] += sum_m g[m, r] * x[m, k]      (g: [M, Rpad] bf16, x: [M, K] bf16, out: [R, K] fp32)
// dA = (s dy B)^T x  and  dB^T = h^T dy  are both of this form.  HBM-bound on x (read once per 8 ranks);
// each workgroup owns 2048 columns x 128 rows and publishes its partial with fp32 atomics (256-byte runs).
// ---------------------------------------------------------------------------
// Layout: a workgroup owns 512 columns (4 waves x the SAME 64 lanes x 8 columns) and `rpb` rows; its 4 waves each take a
// quarter of those rows, their partial sums meet in LDS and ONE wave publishes them.  The fp32 atomics are the cross-
// workgroup part of the reduction and were the bound of the first version (one set per 128 rows: 6.4 M atomics per call at
// 25 200 tokens, ~1 TB/s); with ~32 row groups per call they are 1 M and the kernel streams x.
#define TN_MAXROWS 1024
template <int RC>
__global__ __launch_bounds__(256) void tn_skinny_kernel(const bf16_t* __restrict__ g, const bf16_t* __restrict__ x,
                                                        float* __restrict__ out, int64_t M, int64_t K, int R,
                                                        int Rpad, int64_t ldx, int r0, float scale, int rpb,
                                                        float* __restrict__ part) {
  __shared__ float smem[3 * 64 * RC * 8];                 // 48 KB: first the g rows [rpb][RC], then 3 waves' partials
  static_assert(3 * 64 * RC * 8 >= TN_MAXROWS * RC, "LDS image too small for the g rows");
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t m0 = (int64_t)blockIdx.y * rpb;
  const int64_t k = ((int64_t)blockIdx.x * 64 + lane) * 8;
  const int nrows = (int)((M - m0) < rpb ? (M - m0) : rpb);
  for (int i = threadIdx.x; i < rpb * RC; i += 256) {
    const int m = i / RC, rr = i - m * RC;
    smem[i] = (m < nrows && r0 + rr < R) ? bf2f(g[(m0 + m) * Rpad + r0 + rr]) : 0.f;
  }
  __syncthreads();
  typedef __attribute__((ext_vector_type(2))) float f32x2;
  f32x2 acc[RC][4];                                        // packed pairs of columns: v_pk_fma_f32 (32 per row instead of 64)
#pragma unroll
  for (int rr = 0; rr < RC; ++rr)
#pragma unroll
    for (int e = 0; e < 4; ++e) acc[rr][e] = f32x2{0.f, 0.f};
  const int q = rpb / 4;                                   // rows per wave (rpb % 32 == 0: whole batches of 8, so a batch never
  const int mb = wave * q, me = min(mb + q, nrows);        // reaches into the next wave's rows; rows >= nrows have g = 0 in LDS)
  if (k < K && mb < me) {
    // Two batches of 8 rows in flight: the loads of batch i+1 are issued before the FMAs of batch i (with one wave per
    // SIMD and load -> wait -> compute the kernel sat at ~1 TB/s: nothing was in flight while a wave computed).
    auto load8 = [&](u16x8 (&raw)[8], int m) {
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int mm = m + u < nrows ? m + u : nrows - 1;  // clamped address; its weight in LDS is zero
        raw[u] = *reinterpret_cast<const u16x8*>(x + (m0 + mm) * ldx + k);
      }
    };
    auto fma8 = [&](const u16x8 (&raw)[8], int m) {
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        f32x2 xf[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) xf[e] = f32x2{bf2f(raw[u][2 * e]), bf2f(raw[u][2 * e + 1])};
        static_assert(RC == 8, "the g row is read as two float4");
        const f32x4 g0 = *reinterpret_cast<const f32x4*>(smem + (m + u) * RC);
        const f32x4 g1 = *reinterpret_cast<const f32x4*>(smem + (m + u) * RC + 4);
#pragma unroll
        for (int rr = 0; rr < RC; ++rr) {
          const float gv = rr < 4 ? g0[rr & 3] : g1[rr & 3];
          const f32x2 gvv = f32x2{gv, gv};
#pragma unroll
          for (int e = 0; e < 4; ++e) acc[rr][e] = __builtin_elementwise_fma(gvv, xf[e], acc[rr][e]);
        }
      }
    };
    u16x8 ra[8], rb[8];
    load8(ra, mb);
    for (int m = mb; m < me; m += 16) {
      if (m + 8 < me) load8(rb, m + 8);
      fma8(ra, m);
      if (m + 8 < me) {
        if (m + 16 < me) load8(ra, m + 16);
        fma8(rb, m + 8);
      }
    }
  }
  __syncthreads();                                         // every wave is done with the g rows
  if (wave > 0) {
    float* dst = smem + ((wave - 1) * 64 + lane) * RC * 8;
#pragma unroll
    for (int rr = 0; rr < RC; ++rr)
#pragma unroll
      for (int e = 0; e < 8; ++e) dst[rr * 8 + e] = acc[rr][e >> 1][e & 1];
  }
  __syncthreads();
  if (wave == 0 && k < K) {
#pragma unroll
    for (int rr = 0; rr < RC; ++rr) {
      if (r0 + rr < R) {
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          float v = acc[rr][e >> 1][e & 1];
#pragma unroll
          for (int w2 = 0; w2 < 3; ++w2) v += smem[(w2 * 64 + lane) * RC * 8 + rr * 8 + e];
          if (part) part[((int64_t)blockIdx.y * R + r0 + rr) * K + k + e] = v;   // this row group's slice; summed below
          else atomicAdd(out + (int64_t)(r0 + rr) * K + k + e, v * scale);
        }
      }
    }
  }
}

// out[r, k] = scale * sum over row groups of part[group][r][k]: a fixed order (bit-reproducible, unlike the atomics), no
// zero-fill of `out`, and 64 x fewer L2 atomics - with ~64 row groups the atomics were the larger half of the kernel's time
__global__ __launch_bounds__(256) void tn_skinny_reduce_kernel(const float* __restrict__ part, float* __restrict__ out,
                                                               int64_t n, int groups, float scale) {
  const int64_t i = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4;
  if (i >= n) return;
  f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
  for (int g = 0; g < groups; ++g) acc += *reinterpret_cast<const f32x4*>(part + (int64_t)g * n + i);
  *reinterpret_cast<f32x4*>(out + i) = acc * scale;
}

extern "C" int lcv_tn_skinny(const void* g, const void* x, float* out, int64_t M, int64_t K, int64_t R,
                             int64_t Rpad, int64_t ldx, float scale, float* ws, int64_t ws_bytes, void* stream) {
  LCV_CHECK_ARG(g && x && out, "tn_skinny: null pointer");
  LCV_CHECK_ARG(K % 8 == 0 && ldx % 8 == 0 && R >= 1 && R <= Rpad, "tn_skinny: bad shape");
  if (M == 0) return LCV_OK;
  // ~32 row groups per call; ~64 when there are few column blocks (K <= 4096: 8), so that two workgroups share a CU
  const int64_t groups = (K + 511) / 512 <= 8 ? 64 : 32;
  int64_t rpb = ((M + groups - 1) / groups + 31) / 32 * 32;
  if (rpb > TN_MAXROWS) rpb = TN_MAXROWS;
  if (rpb < 64) rpb = 64;
  const dim3 grid((unsigned)((K + 511) / 512), (unsigned)((M + rpb - 1) / rpb));
  // with a workspace of >= lcv_tn_skinny_ws_bytes(M, K, R): per-group partial sums + a fixed-order reduction (`out` need not
  // be zeroed and is overwritten); without: fp32 atomics into a zero-filled `out`
  const int64_t need = (int64_t)grid.y * R * K * 4;
  float* part = (ws && ws_bytes >= need && ((uintptr_t)ws % 16) == 0) ? ws : nullptr;
  LCV_CHECK_ARG(ws == nullptr || part != nullptr, "tn_skinny: workspace of %ld bytes is too small or misaligned (need %ld)",
                (long)ws_bytes, (long)need);
  for (int r0 = 0; r0 < R; r0 += 8) {
    hipLaunchKernelGGL(tn_skinny_kernel<8>, grid, dim3(256), 0, (hipStream_t)stream, (const bf16_t*)g,
                       (const bf16_t*)x, out, M, K, (int)R, (int)Rpad, ldx, r0, scale, (int)rpb, part);
    LCV_LAUNCH_CHECK("tn_skinny");
  }
  if (part) {
    const int64_t n = R * K;   // K % 8 == 0: whole float4s
    hipLaunchKernelGGL(tn_skinny_reduce_kernel, dim3((unsigned)((n / 4 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, part,
                       out, n, (int)grid.y, scale);
    LCV_LAUNCH_CHECK("tn_skinny_reduce");
  }
  return LCV_OK;
}

extern "C" int64_t lcv_tn_skinny_ws_bytes(int64_t M, int64_t K, int64_t R) {
  const int64_t groups = (K + 511) / 512 <= 8 ? 64 : 32;
  int64_t rpb = ((M + groups - 1) / groups + 31) / 32 * 32;
  if (rpb > TN_MAXROWS) rpb = TN_MAXROWS;
  if (rpb < 64) rpb = 64;
  return ((M + rpb - 1) / rpb) * R * K * 4;
}
