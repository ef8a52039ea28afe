// Interior-tile epilogue of the 16x16x32 GEMM kernels (included by gemm.hip after gemm16_epilogue; used by the 8-phase kernel,
// the four-wave stream kernels and the convolution kernels).
#pragma once

// ---- epilogue of an INTERIOR tile (all 128 rows and 16 TN columns inside the matrix, bf16 output, 8-byte aligned rows) for the
// three epilogues of the DiT's big GEMMs.  With one wave per SIMD nothing runs under the epilogue, so it must be short: the
// generic `gemm16_epilogue` re-derives 64-bit addresses, bounds and alignment per (i, j) and cost 17-30 us per 256 x 256 tile;
// here the bias (and gate) values of the lane's 4 TN columns are loaded once per tile, a row pointer advances by a constant,
// the residual row of block i+1 is requested before block i is finished, and every store is a base + immediate.  Same
// expressions as the generic form, operand for operand: results are bit-identical (tests/test_gpu_kernels.py).  Edge tiles,
// fp32 output, the GELU / SiLU epilogues and a tile whose rows straddle two latent frames (gate rows differ) take the generic one.
// TMB = 16-row blocks per wave (8 in the GEMM kernels, 4 in the row-tile convolution)
template <int EPI, int TN, int TMB = 8>
__device__ __forceinline__ bool g4_fast_epilogue_ok(const GemmParams& p, int64_t mw, int64_t nw) {
  if (p.out_f32 || mw + 16 * TMB > p.M || nw + 16 * TN > p.N || (p.ldc & 3) || ((uintptr_t)p.c & 7)) return false;
  if (p.bias && ((uintptr_t)p.bias & 7)) return false;
  if constexpr (EPI == LCV_EPI_NONE) return true;
  if constexpr (EPI == LCV_EPI_GATE_RESIDUAL) {
    if ((uintptr_t)p.resid & 7) return false;
    if (p.gate && (((uintptr_t)p.gate & 15) || (p.mod_stride & 3) || mw / p.rows_per_frame != (mw + 16 * TMB - 1) / p.rows_per_frame)) return false;
    return true;
  }
  if constexpr (EPI == LCV_EPI_SWIGLU) return (TN % 4 == 0) && (!p.resid || (((uintptr_t)p.resid & 7) == 0 && (p.N & 3) == 0));
  return false;
}

template <int EPI, int TN, int TMB = 8>
__device__ __forceinline__ void g4_fast_epilogue(const GemmParams& p, f32x4v (&acc)[TMB][TN], int64_t mw, int64_t nw, int r16, int q) {
  float bv[TN][4];
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    if (p.bias) {
      const u16x4 b4 = *reinterpret_cast<const u16x4*>(p.bias + nw + 16 * j + 4 * q);
#pragma unroll
      for (int e = 0; e < 4; ++e) bv[j][e] = bf2f(b4[e]);
    } else {
#pragma unroll
      for (int e = 0; e < 4; ++e) bv[j][e] = 0.f;
    }
  }
  if constexpr (EPI == LCV_EPI_NONE) {
    bf16_t* crow = (bf16_t*)p.c + (mw + r16) * p.ldc + nw + 4 * q;
#pragma unroll
    for (int i = 0; i < TMB; ++i) {
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        u16x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = f2bf(p.bias ? acc[i][j][e] + bv[j][e] : acc[i][j][e]);
        *reinterpret_cast<u16x4*>(crow + 16 * j) = o;
      }
      crow += 16 * p.ldc;
      __builtin_amdgcn_sched_barrier(0);
    }
  } else if constexpr (EPI == LCV_EPI_GATE_RESIDUAL) {
    float gv[TN][4];
    if (p.gate) {
      const float* grow = p.gate + (mw / p.rows_per_frame) * p.mod_stride + nw + 4 * q;   // one latent frame for the whole tile
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        const f32x4 g4 = *reinterpret_cast<const f32x4*>(grow + 16 * j);
#pragma unroll
        for (int e = 0; e < 4; ++e) gv[j][e] = g4[e];
      }
    }
    const int64_t off0 = (mw + r16) * p.ldc + nw + 4 * q;
    bf16_t* crow = (bf16_t*)p.c + off0;
    const bf16_t* rrow = p.resid + off0;
    u16x4 rcur[TN], rnxt[TN];
#pragma unroll
    for (int j = 0; j < TN; ++j) rcur[j] = *reinterpret_cast<const u16x4*>(rrow + 16 * j);
#pragma unroll
    for (int i = 0; i < TMB; ++i) {
      if (i < TMB - 1) {
#pragma unroll
        for (int j = 0; j < TN; ++j) rnxt[j] = *reinterpret_cast<const u16x4*>(rrow + 16 * p.ldc + 16 * j);
      }
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        u16x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          float v = acc[i][j][e];
          if (p.bias) v += bv[j][e];
          o[e] = f2bf(bf2f(rcur[j][e]) + (p.gate ? gv[j][e] : 1.0f) * bfround(v));
        }
        *reinterpret_cast<u16x4*>(crow + 16 * j) = o;
      }
#pragma unroll
      for (int j = 0; j < TN; ++j) rcur[j] = rnxt[j];
      crow += 16 * p.ldc;
      rrow += 16 * p.ldc;
      __builtin_amdgcn_sched_barrier(0);
    }
  } else if constexpr (EPI == LCV_EPI_SWIGLU) {
    // W rows interleaved [32 gate | 32 up]: within a 64-column block, 16-wide tiles 0, 1 are gate and 2, 3 their up partners
    bf16_t* crow = (bf16_t*)p.c + (mw + r16) * p.ldc + nw / 2 + 4 * q;
    bf16_t* arow = p.resid ? const_cast<bf16_t*>(p.resid) + (mw + r16) * p.N + nw + 4 * q : nullptr;
#pragma unroll
    for (int i = 0; i < TMB; ++i) {
#pragma unroll
      for (int jb = 0; jb < TN / 4; ++jb)
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          u16x4 o, og, ou;
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const float bg = p.bias ? bv[4 * jb + u][e] : 0.f;
            const float bu = p.bias ? bv[4 * jb + 2 + u][e] : 0.f;
            const float gvv = bfround(acc[i][4 * jb + u][e] + bg);
            const float uvv = bfround(acc[i][4 * jb + 2 + u][e] + bu);
            o[e] = f2bf(bfround(silu_f(gvv)) * uvv);
            og[e] = f2bf(gvv);
            ou[e] = f2bf(uvv);
          }
          *reinterpret_cast<u16x4*>(crow + 32 * jb + 16 * u) = o;
          if (arow) {   // training: the pre-activation (gate | up) rows
            *reinterpret_cast<u16x4*>(arow + 64 * jb + 16 * u) = og;
            *reinterpret_cast<u16x4*>(arow + 64 * jb + 16 * u + 32) = ou;
          }
        }
      crow += 16 * p.ldc;
      if (arow) arow += 16 * p.N;
      __builtin_amdgcn_sched_barrier(0);
    }
  }
}

