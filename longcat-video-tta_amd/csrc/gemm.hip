// Projection GEMM for the DiT linears:  C[M,N] = A[M,K] W[N,K]^T (+ A2[M,K2] W2[N,K2]^T) + bias
// bf16 operands, fp32 accumulation on v_mfma_f32_32x32x16_bf16, fused epilogues.
//
// Structure (gfx950):
//   * BM x BN output tile per workgroup, BK = 64; waves laid out WR x WC, each owning
//     (BM/WR) x (BN/WC) as TM x TN MFMA tiles of 32x32.
//   * both operands are K-contiguous ("NT"): an nn.Linear weight [out,in] is used as stored.
//     The backward dx = dy W uses a resident transposed copy of W (288 GB HBM: 2x27 GB of
//     weights is cheap) so the same kernel serves forward and backward.
//   * global -> LDS by LDS-DMA (global_load_lds_dwordx4, 1 KiB per wave-instruction),
//     two LDS buffers; the LDS image is lane-linear, the XOR swizzle that makes the
//     ds_read_b128 fragment reads conflict-free is applied to the SOURCE address and to
//     the read address (same involution on both sides).
//   * the rank-r LoRA term is one extra 64-deep K-step fed from (a2, w2): zero epilogue cost.
//   * XCD-aware block order: consecutive workgroup ids on one XCD walk down M inside one
//     N panel, so the W panel stays in that XCD's L2.
#include "lcv_common.h"
#include <type_traits>

typedef __attribute__((address_space(3))) void lds_void;
typedef __attribute__((address_space(1))) const void gbl_void;

struct GemmParams {
  const bf16_t* a;
  const bf16_t* w;
  const bf16_t* bias;
  const bf16_t* a2;
  const bf16_t* w2;
  void* c;
  int64_t M, N;
  int nk1, nk2;  // 64-deep K tiles of (a,w) and (a2,w2)
  int64_t lda, ldw, lda2, ldw2, ldc;
  int out_f32;
  const bf16_t* resid;
  const float* gate;  // mod + gate_off
  int64_t rows_per_frame, mod_stride;
  int tiles_m, tiles_n, group_m;
  // tile range of this launch (the persistent 8-phase kernel) and the split-K form of a tail launch: work item
  // blockIdx.x = slice * vid_count + tile_local computes K tiles [slice * nk_split, +nk_split) of tile vid_begin + tile_local
  // and leaves its fp32 accumulators in `ws` (register order); gemm8p_splitk_reduce_kernel adds the slices and runs the epilogue
  int vid_begin, vid_count, splitk, nk_split;
  float* ws;
  int fast_epi;                // 8-phase kernel: interior tiles take gemm_fast_epilogue.h (LCV_GEMM_FAST_EPI=0 turns it off)
  // implicit-GEMM convolution mode (channels-last activations [B,Tin,Hin,Win,Cin], rows m = output pixels):
  // K tiles run over (tap, 64-channel chunk); out-of-range taps read a zero page.
  int cv_T, cv_H, cv_W;        // output extent (rows m = ((b*T + t)*H + h)*W + w)
  int cv_Tin, cv_Hin, cv_Win;  // input extent
  int cv_kt, cv_kh, cv_kw, cv_cpt;  // taps and 64-channel chunks per tap (Cin / 64)
  int cv_up2x;                 // nearest 2x spatial upsample folded into the gather
  int cv_st, cv_sh, cv_sw;     // output -> input strides (downsampling convs of the VAE encoder)
  int cv_pt, cv_ph, cv_pw;     // zero padding IN FRONT of each axis (causal: kt-1 frames; "same": k/2; encoder downsample: 0)
  const bf16_t* cv_zero;       // >= 128 bytes of zeros
};

template <int BM, int BN, int WR, int WC>
struct GemmCfg {
  static constexpr int NW = WR * WC;
  static constexpr int NT = NW * 64;
  static constexpr int TM = BM / WR / 32;
  static constexpr int TN = BN / WC / 32;
  static constexpr int A_BYTES = BM * 128;
  static constexpr int B_BYTES = BN * 128;
  static constexpr int STAGE_BYTES = A_BYTES + B_BYTES;
  static constexpr int IA = BM / 8 / NW;  // LDS-DMA wave-instructions per wave for the A tile
  static constexpr int IB = BN / 8 / NW;
};

__device__ __forceinline__ float gelu_tanh_f(float x) {
  const float k = 0.7978845608028654f;
  const float inner = k * (x + 0.044715f * x * x * x);
  return 0.5f * x * (1.0f + tanhf(inner));
}

// Workgroup id -> output tile.  Ids that share an XCD (id % 8, observed round-robin dealing; speed only) get a contiguous run
// of the tile sequence, and the sequence itself is "grouped": GROUP_M (p.group_m) consecutive tile rows are walked column by
// column, so the ~32 workgroups resident on one XCD cover a GROUP_M x (32 / GROUP_M) block of tiles and share that many A
// and W panels in that XCD's L2 instead of 32 A panels + 1 W panel.
__device__ __forceinline__ void gemm_tile_coords(const GemmParams& p, int orig, int& tm, int& tn) {
  const int GROUP_M = p.group_m;
  const int nwg = p.tiles_m * p.tiles_n;
  const int xcd = orig & 7;
  const int q = nwg >> 3, r8 = nwg & 7;
  const int pid = (xcd < r8 ? xcd * (q + 1) : r8 * (q + 1) + (xcd - r8) * q) + (orig >> 3);
  const int per_group = GROUP_M * p.tiles_n;
  const int group = pid / per_group;
  const int first_m = group * GROUP_M;
  const int gsz = (p.tiles_m - first_m) < GROUP_M ? (p.tiles_m - first_m) : GROUP_M;
  const int in_group = pid - group * per_group;
  tm = first_m + in_group % gsz;
  tn = in_group / gsz;
}
__device__ __forceinline__ void gemm_tile_coords(const GemmParams& p, int& tm, int& tn) {
  gemm_tile_coords(p, (int)blockIdx.x, tm, tn);
}

// ---- shared epilogue ----
// The MFMAs are issued with the WEIGHT fragment as the A operand and the ACTIVATION fragment as the B operand, so the
// accumulator holds C^T: acc[i][j][e] = C[m = mw + 32 i + (lane & 31)][n = nw + 32 j + (e&3) + 8 (e>>2) + 4 (lane>>5)].
// A lane therefore owns ONE output row per i and 4 CONSECUTIVE columns per register group: the store is 8 bytes
// (bf16) or 16 bytes (fp32) per lane and instruction instead of 2 — a quarter of the store instructions, and bias /
// residual / gate come in as 8- and 16-byte loads.
template <int BM, int BN, int WR, int WC, int EPI>
__device__ __forceinline__ void gemm_epilogue(const GemmParams& p,
                                              f32x16 (&acc)[GemmCfg<BM, BN, WR, WC>::TM][GemmCfg<BM, BN, WR, WC>::TN],
                                              int64_t m0, int64_t n0, int wr, int wc, int r, int h) {
  using Cfg = GemmCfg<BM, BN, WR, WC>;
  const int64_t mw = m0 + wr * (BM / WR);
  const int64_t nw = n0 + wc * (BN / WC);
  const bool vec = (p.ldc % 4 == 0) && (((uintptr_t)p.c & 15) == 0);  // 4-column groups are naturally aligned
#pragma unroll
  for (int i = 0; i < Cfg::TM; ++i) {
    const int64_t m = mw + i * 32 + r;
    if (m >= p.M) continue;
    const float* grow = nullptr;
    if constexpr (EPI == LCV_EPI_GATE_RESIDUAL) {
      if (p.gate) grow = p.gate + (m / p.rows_per_frame) * p.mod_stride;
    }
    if constexpr (EPI == LCV_EPI_SWIGLU) {
      // W rows interleaved [32 gate | 32 up]: tile j even = gate, j odd = up of the same 32 features
      bf16_t* C = (bf16_t*)p.c + m * p.ldc;
#pragma unroll
      for (int jj = 0; jj < Cfg::TN / 2; ++jj)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int64_t ng = nw + jj * 64 + 8 * g + 4 * h;  // gate column in the interleaved weight
          if (ng >= p.N) continue;
          const int64_t f = (nw + jj * 64) / 2 + 8 * g + 4 * h;
          u16x4 o;
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const float bg = p.bias ? bf2f(p.bias[ng + e]) : 0.f;
            const float bu = p.bias ? bf2f(p.bias[ng + 32 + e]) : 0.f;
            const float gv = bfround(acc[i][2 * jj][4 * g + e] + bg);
            const float uv = bfround(acc[i][2 * jj + 1][4 * g + e] + bu);
            o[e] = f2bf(bfround(silu_f(gv)) * uv);
            if (p.resid) {   // training: the pre-activation [gate | up] row, in the interleaved weight's own column order
              bf16_t* aux = const_cast<bf16_t*>(p.resid) + m * p.N;
              aux[ng + e] = f2bf(gv);
              aux[ng + 32 + e] = f2bf(uv);
            }
          }
          if (vec) *reinterpret_cast<u16x4*>(C + f) = o;
          else {
#pragma unroll
            for (int e = 0; e < 4; ++e) C[f + e] = o[e];
          }
        }
    } else {
#pragma unroll
      for (int j = 0; j < Cfg::TN; ++j)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int64_t n = nw + j * 32 + 8 * g + 4 * h;
          if (n >= p.N) continue;
          const bool full = vec && (n + 3 < p.N);
          float v[4];
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = acc[i][j][4 * g + e];
          if (p.bias) {
            if (full) {
              const u16x4 b4 = *reinterpret_cast<const u16x4*>(p.bias + n);
#pragma unroll
              for (int e = 0; e < 4; ++e) v[e] += bf2f(b4[e]);
            } else {
#pragma unroll
              for (int e = 0; e < 4; ++e)
                if (n + e < p.N) v[e] += bf2f(p.bias[n + e]);
            }
          }
          if constexpr (EPI == LCV_EPI_GATE_RESIDUAL) {
            // the projection output is a bf16 tensor upstream: round, then resid + gate * xs in fp32
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              if (n + e < p.N) {
                const float gt = grow ? grow[n + e] : 1.0f;
                v[e] = bf2f(p.resid[m * p.ldc + n + e]) + gt * bfround(v[e]);
              }
            }
          } else if constexpr (EPI == LCV_EPI_GELU_TANH) {
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = gelu_tanh_f(bfround(v[e]));
          } else if constexpr (EPI == LCV_EPI_SILU) {
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = silu_f(bfround(v[e]));
          }
          if (p.out_f32) {
            float* C = (float*)p.c + m * p.ldc + n;
            if (full) *reinterpret_cast<f32x4*>(C) = f32x4{v[0], v[1], v[2], v[3]};
            else {
#pragma unroll
              for (int e = 0; e < 4; ++e)
                if (n + e < p.N) C[e] = v[e];
            }
          } else {
            bf16_t* C = (bf16_t*)p.c + m * p.ldc + n;
            if (full) {
              u16x4 o;
#pragma unroll
              for (int e = 0; e < 4; ++e) o[e] = f2bf(v[e]);
              *reinterpret_cast<u16x4*>(C) = o;
            } else {
#pragma unroll
              for (int e = 0; e < 4; ++e)
                if (n + e < p.N) C[e] = f2bf(v[e]);
            }
          }
        }
    }
  }
}

template <int BM, int BN, int WR, int WC, int EPI, bool CONV>
__global__ __launch_bounds__(WR* WC * 64) void gemm_nt_kernel(const GemmParams p) {
  using Cfg = GemmCfg<BM, BN, WR, WC>;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

  int tm, tn;
  gemm_tile_coords(p, tm, tn);
  const int64_t m0 = (int64_t)tm * BM;
  const int64_t n0 = (int64_t)tn * BN;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave / WC, wc = wave % WC;
  const int r = lane & 31, h = lane >> 5;

  // ---- per-lane source rows for the LDS-DMA (clamped: tails re-read the last valid row) ----
  const int ld_row = lane >> 3, ld_slot = lane & 7;
  int64_t a_row[Cfg::IA], b_row[Cfg::IB];
  int a_sw[Cfg::IA], b_sw[Cfg::IB];
  int cv_t[Cfg::IA], cv_h[Cfg::IA], cv_w[Cfg::IA];  // conv mode: output pixel of each staged row
#pragma unroll
  for (int t = 0; t < Cfg::IA; ++t) {
    const int row = (wave * Cfg::IA + t) * 8 + ld_row;
    int64_t g = m0 + row;
    if (g > p.M - 1) g = p.M - 1;
    a_row[t] = g;
    a_sw[t] = (ld_slot ^ ((row >> 1) & 7)) * 8;
    if constexpr (CONV) {
      int64_t r2 = g;
      cv_w[t] = (int)(r2 % p.cv_W); r2 /= p.cv_W;
      cv_h[t] = (int)(r2 % p.cv_H); r2 /= p.cv_H;
      cv_t[t] = (int)(r2 % p.cv_T);
      a_row[t] = r2 / p.cv_T;  // batch index
    }
  }
#pragma unroll
  for (int t = 0; t < Cfg::IB; ++t) {
    const int row = (wave * Cfg::IB + t) * 8 + ld_row;
    int64_t g = n0 + row;
    if (g > p.N - 1) g = p.N - 1;
    b_row[t] = g;
    b_sw[t] = (ld_slot ^ ((row >> 1) & 7)) * 8;
  }

  auto stage = [&](int kt, int buf) {
    const bf16_t* A = p.a;
    const bf16_t* W = p.w;
    int64_t lda = p.lda, ldw = p.ldw;
    int k0 = kt * 64;
    if (kt >= p.nk1) {
      A = p.a2; W = p.w2; lda = p.lda2; ldw = p.ldw2; k0 = (kt - p.nk1) * 64;
    }
    unsigned char* sa = smem + buf * Cfg::STAGE_BYTES;
    unsigned char* sb = sa + Cfg::A_BYTES;
    if constexpr (CONV) {
      const int tap = kt / p.cv_cpt;
      const int c0 = (kt - tap * p.cv_cpt) * 64;
      const int dw = tap % p.cv_kw;
      const int dh = (tap / p.cv_kw) % p.cv_kh;
      const int dt = tap / (p.cv_kw * p.cv_kh);
#pragma unroll
      for (int t = 0; t < Cfg::IA; ++t) {
        const int ti = cv_t[t] * p.cv_st + dt - p.cv_pt;  // front padding only: causal in t, (k/2 | 0) in h, w
        int hi = cv_h[t] * p.cv_sh + dh - p.cv_ph;
        int wi = cv_w[t] * p.cv_sw + dw - p.cv_pw;
        const int hb = p.cv_up2x ? 2 * p.cv_Hin : p.cv_Hin, wb = p.cv_up2x ? 2 * p.cv_Win : p.cv_Win;
        const bool ok = ti >= 0 && ti < p.cv_Tin && hi >= 0 && hi < hb && wi >= 0 && wi < wb;
        if (p.cv_up2x) { hi >>= 1; wi >>= 1; }
        const int64_t pix = ((a_row[t] * p.cv_Tin + ti) * p.cv_Hin + hi) * (int64_t)p.cv_Win + wi;
        const bf16_t* src = ok ? (A + pix * lda + c0 + a_sw[t]) : (p.cv_zero + a_sw[t]);
        __builtin_amdgcn_global_load_lds((gbl_void*)src, (lds_void*)(sa + (wave * Cfg::IA + t) * 1024), 16, 0, 0);
      }
    } else {
#pragma unroll
      for (int t = 0; t < Cfg::IA; ++t) {
        const bf16_t* src = A + a_row[t] * lda + k0 + a_sw[t];
        __builtin_amdgcn_global_load_lds((gbl_void*)src, (lds_void*)(sa + (wave * Cfg::IA + t) * 1024), 16, 0, 0);
      }
    }
#pragma unroll
    for (int t = 0; t < Cfg::IB; ++t) {
      const bf16_t* src = W + b_row[t] * ldw + k0 + b_sw[t];
      __builtin_amdgcn_global_load_lds((gbl_void*)src, (lds_void*)(sb + (wave * Cfg::IB + t) * 1024), 16, 0, 0);
    }
  };

  f32x16 acc[Cfg::TM][Cfg::TN];
#pragma unroll
  for (int i = 0; i < Cfg::TM; ++i)
#pragma unroll
    for (int j = 0; j < Cfg::TN; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  const int sw = (r >> 1) & 7;
  const int a_off = (wr * (BM / WR) + r) * 128;
  const int b_off = (wc * (BN / WC) + r) * 128;
  const int nk = p.nk1 + p.nk2;

  stage(0, 0);
  for (int kt = 0; kt < nk; ++kt) {
    __syncthreads();  // drains this wave's LDS-DMA (vmcnt 0) and orders buffer reuse
    if (kt + 1 < nk) stage(kt + 1, (kt + 1) & 1);
    const unsigned char* sa = smem + (kt & 1) * Cfg::STAGE_BYTES;
    const unsigned char* sb = sa + Cfg::A_BYTES;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      const int ch = ((2 * ks + h) ^ sw) * 16;
      bf16x8 af[Cfg::TM], bfr[Cfg::TN];
#pragma unroll
      for (int i = 0; i < Cfg::TM; ++i)
        af[i] = *reinterpret_cast<const bf16x8*>(sa + a_off + i * 32 * 128 + ch);
#pragma unroll
      for (int j = 0; j < Cfg::TN; ++j)
        bfr[j] = *reinterpret_cast<const bf16x8*>(sb + b_off + j * 32 * 128 + ch);
#pragma unroll
      for (int i = 0; i < Cfg::TM; ++i)
#pragma unroll
        for (int j = 0; j < Cfg::TN; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bfr[j], af[i], acc[i][j], 0, 0, 0);  // C^T tile
    }
  }

  gemm_epilogue<BM, BN, WR, WC, EPI>(p, acc, m0, n0, wr, wc, r, h);
}

// ---------------------------------------------------------------------------
// v_mfma_f32_16x16x32_bf16 variant of the two-buffer kernel (same tiles, LDS image, DMA and tile order).
// Why: both MFMA shapes cost the same cycles per flop, but on random data the chip holds a higher clock on the 16x16x32
// shape (MI355X_MICROARCH.md, DVFS give-back item 7: ~1.12-1.15x flop/s with operands re-read from LDS), and these
// kernels are power/clock-limited, not issue-limited (every re-schedule of the 32x32x16 loop landed at the same ~1.0 PF/s).
// Fragment maps: A-op lane l = W[16 j + (l & 15)][32 s + 8 (l >> 4) + 0..7], B-op lane l = X[16 i + (l & 15)][same k];
// accumulator (C^T): acc[i][j][e] = C[m = 16 i + (l & 15)][n = 16 j + 4 (l >> 4) + e]  -> 8-byte stores again.
// The (row >> 1) & 7 XOR swizzle of the 128-byte-row LDS image is conflict-free for these reads as well: a
// ds_read_b128 lane group mixes rows of chunk c (even) and c + 1 = c ^ 1, whose swizzled slots stay disjoint.
// ---------------------------------------------------------------------------
typedef __attribute__((ext_vector_type(4))) float f32x4v;

// ---- epilogue of the 16x16x32 kernels: acc[i][j][e] = C[m = mw + 16 i + (lane & 15)][n = nw + 16 j + 4 (lane >> 4) + e] ----
// ROW_FENCE: a scheduling fence after every 16-row block, for kernels whose accumulators live in AGPRs - without it hipcc
// hoists all TM x TN accumulator reads (and the address arithmetic of every store) to the top and spills around them.
// PAIRED (gemm4k.h): the weight rows of a 32-column block are dealt to its two 16-row MFMA tiles so that a lane's 4 + 4 values of
// tiles (2u, 2u + 1) are 8 CONSECUTIVE columns: acc[i][j][e] = C[m][n = nw + 32 (j >> 1) + 8 (lane >> 4) + 4 (j & 1) + e].
template <int TM, int TN, int EPI, bool ROW_FENCE = false, bool PAIRED = false>
__device__ __forceinline__ void gemm16_epilogue(const GemmParams& p, f32x4v (&acc)[TM][TN], int64_t mw, int64_t nw, int r16,
                                                int q) {
  // ---- epilogue: lane owns row m and 4 consecutive columns per (i, j) ----
  const bool vec = (p.ldc % 4 == 0) && (((uintptr_t)p.c & 15) == 0);
  const bool res_vec = EPI == LCV_EPI_GATE_RESIDUAL && (((uintptr_t)p.resid & 7) == 0);   // rows are ldc apart, ldc % 4 == 0 under vec
#pragma unroll
  for (int i = 0; i < TM; ++i) {
    if constexpr (ROW_FENCE) __builtin_amdgcn_sched_barrier(0);
    const int64_t m = mw + i * 16 + r16;
    if (m >= p.M) continue;
    const float* grow = nullptr;
    if constexpr (EPI == LCV_EPI_GATE_RESIDUAL) {
      if (p.gate) grow = p.gate + (m / p.rows_per_frame) * p.mod_stride;
    }
    if constexpr (EPI == LCV_EPI_SWIGLU) {
      // W rows interleaved [32 gate | 32 up]: within a 64-row block, 16-wide tiles 0,1 are gate and 2,3 their up partners
      bf16_t* C = (bf16_t*)p.c + m * p.ldc;
#pragma unroll
      for (int jb = 0; jb < TN / 4; ++jb)
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          const int in_blk = PAIRED ? 8 * q + 4 * u : 16 * u + 4 * q;
          const int64_t ng = nw + jb * 64 + in_blk;
          if (ng >= p.N) continue;
          const int64_t f = (nw + jb * 64) / 2 + in_blk;
          u16x4 o, og, ou;
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const float bg = p.bias ? bf2f(p.bias[ng + e]) : 0.f;
            const float bu = p.bias ? bf2f(p.bias[ng + 32 + e]) : 0.f;
            const float gv = bfround(acc[i][4 * jb + u][e] + bg);
            const float uv = bfround(acc[i][4 * jb + 2 + u][e] + bu);
            o[e] = f2bf(bfround(silu_f(gv)) * uv);
            og[e] = f2bf(gv);
            ou[e] = f2bf(uv);
          }
          if (p.resid) {   // training: the pre-activation [gate | up] row, in the interleaved weight's own column order
            bf16_t* aux = const_cast<bf16_t*>(p.resid) + m * p.N;
            if (p.N % 4 == 0 && (((uintptr_t)p.resid & 7) == 0)) {
              *reinterpret_cast<u16x4*>(aux + ng) = og;
              *reinterpret_cast<u16x4*>(aux + ng + 32) = ou;
            } else {
#pragma unroll
              for (int e = 0; e < 4; ++e) { aux[ng + e] = og[e]; aux[ng + 32 + e] = ou[e]; }
            }
          }
          if (vec) *reinterpret_cast<u16x4*>(C + f) = o;
          else {
#pragma unroll
            for (int e = 0; e < 4; ++e) C[f + e] = o[e];
          }
        }
    } else {
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        const int64_t n = nw + (PAIRED ? (j >> 1) * 32 + 8 * q + 4 * (j & 1) : j * 16 + 4 * q);
        if (n >= p.N) continue;
        const bool full = vec && (n + 3 < p.N);
        float v[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = acc[i][j][e];
        if (p.bias) {
          if (full) {
            const u16x4 b4 = *reinterpret_cast<const u16x4*>(p.bias + n);
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] += bf2f(b4[e]);
          } else {
#pragma unroll
            for (int e = 0; e < 4; ++e)
              if (n + e < p.N) v[e] += bf2f(p.bias[n + e]);
          }
        }
        if constexpr (EPI == LCV_EPI_GATE_RESIDUAL) {
          if (full && res_vec) {   // one 8-byte residual load (and one 16-byte gate load) per 4 columns instead of 4 + 4 scalar ones
            const u16x4 r4 = *reinterpret_cast<const u16x4*>(p.resid + m * p.ldc + n);
            float g4[4] = {1.0f, 1.0f, 1.0f, 1.0f};
            if (grow) {
              if ((((uintptr_t)(grow + n)) & 15) == 0) {
                const f32x4v gv = *reinterpret_cast<const f32x4v*>(grow + n);
#pragma unroll
                for (int e = 0; e < 4; ++e) g4[e] = gv[e];
              } else {
#pragma unroll
                for (int e = 0; e < 4; ++e) g4[e] = grow[n + e];
              }
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = bf2f(r4[e]) + g4[e] * bfround(v[e]);
          } else {
#pragma unroll
            for (int e = 0; e < 4; ++e)
              if (n + e < p.N) v[e] = bf2f(p.resid[m * p.ldc + n + e]) + (grow ? grow[n + e] : 1.0f) * bfround(v[e]);
          }
        } else if constexpr (EPI == LCV_EPI_GELU_TANH) {
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = gelu_tanh_f(bfround(v[e]));
        } else if constexpr (EPI == LCV_EPI_SILU) {
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = silu_f(bfround(v[e]));
        }
        if (p.out_f32) {
          float* C = (float*)p.c + m * p.ldc + n;
          if (full) *reinterpret_cast<f32x4*>(C) = f32x4{v[0], v[1], v[2], v[3]};
          else {
#pragma unroll
            for (int e = 0; e < 4; ++e)
              if (n + e < p.N) C[e] = v[e];
          }
        } else {
          bf16_t* C = (bf16_t*)p.c + m * p.ldc + n;
          if (full) {
            u16x4 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = f2bf(v[e]);
            *reinterpret_cast<u16x4*>(C) = o;
          } else {
#pragma unroll
            for (int e = 0; e < 4; ++e)
              if (n + e < p.N) C[e] = f2bf(v[e]);
          }
        }
      }
    }
  }
}

// NST = 3: a ring of three K-tile buffers, the LDS-DMA of K tile kt + 2 issued at the top of K tile kt and a COUNTED wait (the
// newest tile may stay in flight across the barrier).  With two buffers the only request in flight is the one issued one
// K tile earlier.  Tried on the VAE's wide convolutions (1.7 us per K tile for 0.7 us of MFMA work): no gain, so the round trip
// is not what they wait for (profiles/r02_conv_rows.md); opt-in (LCV_CONV_N192=3).  Same products in the same order:
// bit-identical to NST = 2.
template <int BM, int BN, int WR, int WC, int EPI, bool CONV, int NST = 2>
__global__ __launch_bounds__(WR* WC * 64) void gemm16_nt_kernel(const GemmParams p) {
  using Cfg = GemmCfg<BM, BN, WR, WC>;
  constexpr int TM = BM / WR / 16, TN = BN / WC / 16;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  int tm, tn;
  gemm_tile_coords(p, tm, tn);
  const int64_t m0 = (int64_t)tm * BM;
  const int64_t n0 = (int64_t)tn * BN;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave / WC, wc = wave % WC;
  const int r16 = lane & 15, q = lane >> 4;

  const int ld_row = lane >> 3, ld_slot = lane & 7;
  int64_t a_row[Cfg::IA], b_row[Cfg::IB];
  int a_sw[Cfg::IA], b_sw[Cfg::IB];
  // conv mode, per staged row and once per output tile: the input pixel index of tap (0, 0, 0) - possibly outside the tensor -,
  // one bit per tap saying whether that tap reads inside it (else: the zero page) and, for the folded 2x upsample, the parity
  // of the upsampled row / column.  A K tile then costs one add, one bit test and one 64-bit multiply-add per row.  Deriving
  // (t, h, w), the bounds and the pixel index anew for every K tile took ~130 vector instructions per 64 MFMAs, several of them
  // quarter-rate integer multiplies: the convolution kernels were bound by address arithmetic, not by the matrix pipe.
  int cv_pix0[Cfg::IA], a_sw2[Cfg::IA];
  unsigned cv_ok[Cfg::IA], cv_par[Cfg::IA];
  const bf16_t* b_ptr[Cfg::IB];   // conv mode: one K extent, so the weight row pointers are fixed
#pragma unroll
  for (int t = 0; t < Cfg::IA; ++t) {
    const int row = (wave * Cfg::IA + t) * 8 + ld_row;
    int64_t g = m0 + row;
    a_row[t] = g > p.M - 1 ? p.M - 1 : g;
    a_sw[t] = (ld_slot ^ ((row >> 1) & 7)) * 8;
    a_sw2[t] = a_sw[t] * 2;
    if constexpr (CONV) {
      int64_t r2 = a_row[t];
      const int wo = (int)(r2 % p.cv_W); r2 /= p.cv_W;
      const int ho = (int)(r2 % p.cv_H); r2 /= p.cv_H;
      const int to = (int)(r2 % p.cv_T);
      const int bb = (int)(r2 / p.cv_T);  // batch index
      // front padding only: causal in t, (k/2 | 0) in h, w
      const int t0 = to * p.cv_st - p.cv_pt, h0 = ho * p.cv_sh - p.cv_ph, w0 = wo * p.cv_sw - p.cv_pw;
      const int hb = p.cv_up2x ? 2 * p.cv_Hin : p.cv_Hin, wb = p.cv_up2x ? 2 * p.cv_Win : p.cv_Win;
      unsigned ok = 0;
      int tap = 0;
      for (int dt = 0; dt < p.cv_kt; ++dt)
        for (int dh = 0; dh < p.cv_kh; ++dh)
          for (int dw = 0; dw < p.cv_kw; ++dw, ++tap) {
            const int ti = t0 + dt, hi = h0 + dh, wi = w0 + dw;
            if (ti >= 0 && ti < p.cv_Tin && hi >= 0 && hi < hb && wi >= 0 && wi < wb) ok |= 1u << tap;
          }
      cv_ok[t] = ok;
      // with the upsample: floor((x + d) / 2) = (x >> 1) + (((x & 1) + d) >> 1) for d >= 0 (arithmetic shift, x may be -1)
      const int hq = p.cv_up2x ? (h0 >> 1) : h0, wq = p.cv_up2x ? (w0 >> 1) : w0;
      cv_par[t] = p.cv_up2x ? (unsigned)((h0 & 1) | ((w0 & 1) << 1)) : 0u;
      cv_pix0[t] = ((bb * p.cv_Tin + t0) * p.cv_Hin + hq) * p.cv_Win + wq;  // |.| < 2^31: the host checks the pixel count
    }
  }
#pragma unroll
  for (int t = 0; t < Cfg::IB; ++t) {
    const int row = (wave * Cfg::IB + t) * 8 + ld_row;
    int64_t g = n0 + row;
    b_row[t] = g > p.N - 1 ? p.N - 1 : g;
    b_sw[t] = (ld_slot ^ ((row >> 1) & 7)) * 8;
    b_ptr[t] = p.w + b_row[t] * p.ldw + b_sw[t];
  }
  auto stage = [&](int kt, int buf) {
    const bf16_t* A = p.a;
    const bf16_t* W = p.w;
    int64_t lda = p.lda, ldw = p.ldw;
    int k0 = kt * 64;
    if (kt >= p.nk1) { A = p.a2; W = p.w2; lda = p.lda2; ldw = p.ldw2; k0 = (kt - p.nk1) * 64; }
    unsigned char* sa = smem + buf * Cfg::STAGE_BYTES;
    unsigned char* sb = sa + Cfg::A_BYTES;
    if constexpr (CONV) {  // implicit GEMM: K tiles run over (tap, 64-channel chunk); padding taps read the zero page
      const int tap = kt / p.cv_cpt;
      const int c0 = (kt - tap * p.cv_cpt) * 64;
      const int dw = tap % p.cv_kw;
      const int dh = (tap / p.cv_kw) % p.cv_kh;
      const int dt = tap / (p.cv_kw * p.cv_kh);
      const int dpix_t = dt * p.cv_Hin * p.cv_Win;               // all scalar
      const int dpix = dpix_t + dh * p.cv_Win + dw;
      const unsigned row_bytes = (unsigned)p.lda * 2u;            // channels-last row of one pixel
      const uint64_t abase = (uint64_t)(p.a + c0), zbase = (uint64_t)p.cv_zero;
#pragma unroll
      for (int t = 0; t < Cfg::IA; ++t) {
        int pix;
        if (p.cv_up2x) {
          const int ih = (int)((cv_par[t] & 1u) + (unsigned)dh) >> 1, iw = (int)((cv_par[t] >> 1) + (unsigned)dw) >> 1;
          pix = cv_pix0[t] + dpix_t + __mul24(ih, p.cv_Win) + iw;
        } else {
          pix = cv_pix0[t] + dpix;
        }
        // selects, not a branch: both sides are cheap and a divergent branch per row would serialise the four DMAs
        const bool ok = (cv_ok[t] >> tap) & 1u;
        const uint64_t base = ok ? abase : zbase;
        const unsigned pm = ok ? (unsigned)pix : 0u;
        const char* src = (const char*)(base + (uint64_t)pm * row_bytes + (unsigned)a_sw2[t]);
        __builtin_amdgcn_global_load_lds((gbl_void*)src, (lds_void*)(sa + (wave * Cfg::IA + t) * 1024), 16, 0, 0);
      }
    } else {
#pragma unroll
      for (int t = 0; t < Cfg::IA; ++t)
        __builtin_amdgcn_global_load_lds((gbl_void*)(A + a_row[t] * lda + k0 + a_sw[t]),
                                         (lds_void*)(sa + (wave * Cfg::IA + t) * 1024), 16, 0, 0);
    }
    if constexpr (CONV) {
#pragma unroll
      for (int t = 0; t < Cfg::IB; ++t)
        __builtin_amdgcn_global_load_lds((gbl_void*)(b_ptr[t] + k0), (lds_void*)(sb + (wave * Cfg::IB + t) * 1024), 16, 0, 0);
    } else {
#pragma unroll
      for (int t = 0; t < Cfg::IB; ++t)
        __builtin_amdgcn_global_load_lds((gbl_void*)(W + b_row[t] * ldw + k0 + b_sw[t]),
                                         (lds_void*)(sb + (wave * Cfg::IB + t) * 1024), 16, 0, 0);
    }
  };

  f32x4v acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j) acc[i][j] = f32x4v{0.f, 0.f, 0.f, 0.f};

  const int sw = (r16 >> 1) & 7;
  const int a_off = (wr * (BM / WR) + r16) * 128;
  const int b_off = (wc * (BN / WC) + r16) * 128;
  const int nk = p.nk1 + p.nk2;
  stage(0, 0);
  if constexpr (NST == 3) { if (nk > 1) stage(1, 1); }
  int cbuf = 0, sbuf = 2;                         // NST = 3: buffer being multiplied / staged next
  for (int kt = 0; kt < nk; ++kt) {
    if constexpr (NST == 3) {
      // every wave issues IA + IB requests per K tile, so "all but the newest tile" is a compile-time count
      if (kt + 1 < nk) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(Cfg::IA + Cfg::IB) : "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();               // K tile kt has landed for everyone; everyone is done reading buffer sbuf (K tile kt - 1)
      if (kt + 2 < nk) stage(kt + 2, sbuf);
    } else {
      __syncthreads();
      if (kt + 1 < nk) stage(kt + 1, (kt + 1) & 1);
    }
    const unsigned char* sa = smem + (NST == 3 ? cbuf : (kt & 1)) * Cfg::STAGE_BYTES;
    const unsigned char* sb = sa + Cfg::A_BYTES;
    if constexpr (NST == 3) { cbuf = cbuf == 2 ? 0 : cbuf + 1; sbuf = sbuf == 2 ? 0 : sbuf + 1; }
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      const int ch = ((4 * ks + q) ^ sw) * 16;
      bf16x8 af[TM], bfr[TN];
#pragma unroll
      for (int i = 0; i < TM; ++i) af[i] = *reinterpret_cast<const bf16x8*>(sa + a_off + i * 16 * 128 + ch);
#pragma unroll
      for (int j = 0; j < TN; ++j) bfr[j] = *reinterpret_cast<const bf16x8*>(sb + b_off + j * 16 * 128 + ch);
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[j], af[i], acc[i][j], 0, 0, 0);  // C^T tile
    }
  }

  gemm16_epilogue<TM, TN, EPI>(p, acc, m0 + wr * (BM / WR), n0 + wc * (BN / WC), r16, q);
}

// ---------------------------------------------------------------------------
// 256 x 256 tile, 8 phases per pair of K tiles ("ping-pong"): the default for the big token-side projections.
//
// The kernels above issue {stage next tile; read fragments; 64 MFMAs; vmcnt(0) + barrier} per K tile: both waves of a SIMD
// read LDS at the same time and then fight for the matrix core at the same time, and every tile boundary drains the
// LDS-DMA queue.  Here (cdna_hip_programming.md, "The 256^2 8-phase template"):
//   * a K tile is FOUR phases of {ds_read one fragment sub-tile; issue one half-tile of LDS-DMA; counted vmcnt; s_barrier;
//     16 MFMAs = one 64 x 32 quadrant of the wave's 128 x 64 output; s_barrier};
//   * the lower wave row (wr = 1, waves 4-7: the second wave of every SIMD) runs ONE BARRIER BEHIND the upper one, so on
//     each SIMD one wave's MFMA cluster always overlaps the other wave's LDS reads and DMA issue;
//   * the DMA queue is never drained inside the loop: `s_waitcnt vmcnt(8)` leaves the four newest half-tiles in flight
//     across the barriers (raw s_barrier: __syncthreads() would add vmcnt(0)).
// LDS image (128 KiB): 2 K-tile buffers x 4 slots of 128 rows x 128 B, a slot = what one phase of all 8 waves reads:
//     A-mq: rows {64 mq .. +64} of both 128-row wave rows      W-nq: columns {32 nq .. +32} of all four 64-column wave columns
// Hazards (phases counted per wave row; P1..P4 of K tile kt, buffer kt & 1):
//     reads   P1: A-mq0, W-nq0   P2: W-nq1   P3: A-mq1   P4: none (W-nq0 is still in registers)
//     stages  P1: W-nq1(kt+1)    P2: A-mq1(kt+1)    P3: A-mq0(kt+2)    P4: W-nq0(kt+2)
//   WAR: a slot is restaged two or more phases after the phase that read it (both wave rows have retired those reads by
//        then: lgkmcnt(0) follows the first barrier of the reading phase, the staggered row is one barrier later).
//   RAW: the wait of phase p (before its first barrier) retires everything but the 4 newest half-tiles, which always
//        includes every slot phase p+1 reads; the reader passes at least one more barrier than any waiter.
// ---------------------------------------------------------------------------
#define GSTAMP() do {} while (0)

#include "gemm_fast_epilogue.h"

template <int N>
__device__ __forceinline__ void wait_vmcnt() {
  if constexpr (N == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  else if constexpr (N == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
  else if constexpr (N == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
  else if constexpr (N == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
  else if constexpr (N == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
}

// PERSIST: one workgroup per CU walks the tile sequence (virtual id = blockIdx.x + i * gridDim.x: gridDim.x is a multiple of 8,
// so a workgroup keeps its XCD's run of the sequence), and the LDS-DMA prologue of the NEXT tile is issued before the
// epilogue of the current one - the 128 KiB LDS image allows one workgroup per CU, so without this every tile exposes
// its own pipeline fill and its store tail.
// CONV: the A operand is the implicit-GEMM gather of a causal convolution (conv3d_impl): K tile kts = (tap, 64-channel chunk),
// row m = output pixel.  Per staged row and output tile the kernel keeps the pixel index of tap (0, 0, 0), one validity bit
// per tap and the parity bits of the folded upsample; per K tile the tap's coordinates come from a small LDS table behind
// the two K-tile buffers (filled once per workgroup, the divisions 256 wide) - no index arithmetic in the phase loop.
template <int EPI, bool PERSIST, bool SPLIT = false, bool CONV = false>
__global__ __launch_bounds__(512) void gemm8p_nt_kernel(const GemmParams p) {
  constexpr int BUF_BYTES = 65536, SLOT_BYTES = 16384;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 2, wc = wave & 3;
  const int r16 = lane & 15, q = lane >> 4;
  const int nwg = p.vid_begin + p.vid_count;   // end of this launch's tile range
  const bool fast_epi = !CONV && p.fast_epi != 0;
  int k_off = 0;                               // first K tile of this work item (split-K tail launches)
  int nk = p.nk1 + p.nk2;                      // >= 2, nk1 >= 2 (host-checked)
  if constexpr (SPLIT) {
    k_off = (int)(blockIdx.x / (unsigned)p.vid_count) * p.nk_split;
    nk = min(p.nk_split, nk - k_off);          // >= 2 (host-checked)
  }

  // ---- LDS-DMA roles: instruction t of wave w fills slot rows 8 (2 w + t) .. +8; lane -> row (lane >> 3), 16-B position
  // (lane & 7) which holds logical chunk (lane & 7) ^ ((row >> 1) & 7).  Per-lane state is the (clamped) global row of
  // each of the 8 instructions; the source is  uniform base + K offset (SGPRs)  +  row * row-bytes + swizzle (32-bit VGPR) ----
  int arow[2][2], wrow[2][2];  // [mq | nq][t]
  unsigned cv_ok[2][2], cv_par[2][2];   // CONV: arow = pixel index of tap (0, 0, 0); validity bit per tap; upsample parities
  unsigned swz[2];
  // CONV: the K-tile cursor of each A stream (mq = 0, 1): both are staged for K tiles 0, 1, 2, ... of a tile in order, so the
  // tap coordinates advance by scalar increments (reset by the prologue) instead of being derived from kts by divisions
  int cvc_chunk[2] = {0, 0}, cvc_dw[2] = {0, 0}, cvc_dh[2] = {0, 0}, cvc_dt[2] = {0, 0}, cvc_tap[2] = {0, 0};
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    const int srow = 8 * (2 * wave + t) + (lane >> 3);
    swz[t] = (unsigned)(((lane & 7) ^ ((srow >> 1) & 7)) * 16);
  }
  int64_t m0 = 0, n0 = 0;
  // byte offsets of this tile's first row in each operand (wave-uniform, 64-bit, SGPRs): the per-lane part of a source address is
  // then (row inside the tile) * row-bytes + swizzle <= 255 * 2 * ld + 112, which fits 32 bits for ANY M (the literal 49x90x160
  // latent config runs M = 352 800 rows of 22 016 bytes through w2: 7.8e9 bytes from the operand base)
  int64_t a_tile = 0, a2_tile = 0, w_tile = 0, w2_tile = 0;
  auto setup_tile = [&](int vid) {
    int tm, tn;
    gemm_tile_coords(p, vid, tm, tn);
    m0 = (int64_t)tm * 256;
    n0 = (int64_t)tn * 256;
    if constexpr (!CONV) { a_tile = m0 * p.lda * 2; a2_tile = m0 * p.lda2 * 2; }
    w_tile = n0 * p.ldw * 2; w2_tile = n0 * p.ldw2 * 2;
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      const int srow = 8 * (2 * wave + t) + (lane >> 3);
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        int64_t g = m0 + (srow >> 6) * 128 + h * 64 + (srow & 63);
        g = g > p.M - 1 ? p.M - 1 : g;
        arow[h][t] = CONV ? (int)g : (int)(g - m0);        // CONV: replaced below by the pixel index of tap (0, 0, 0)
        if constexpr (CONV) {
          int64_t r2 = g;
          const int wo = (int)(r2 % p.cv_W); r2 /= p.cv_W;
          const int ho = (int)(r2 % p.cv_H); r2 /= p.cv_H;
          const int to = (int)(r2 % p.cv_T);
          const int bb = (int)(r2 / p.cv_T);
          const int t0 = to * p.cv_st - p.cv_pt, h0 = ho * p.cv_sh - p.cv_ph, w0 = wo * p.cv_sw - p.cv_pw;
          const int hb = p.cv_up2x ? 2 * p.cv_Hin : p.cv_Hin, wb = p.cv_up2x ? 2 * p.cv_Win : p.cv_Win;
          unsigned ok = 0;
          int tap = 0;
          for (int dt = 0; dt < p.cv_kt; ++dt)
            for (int dh = 0; dh < p.cv_kh; ++dh)
              for (int dw = 0; dw < p.cv_kw; ++dw, ++tap) {
                const int ti = t0 + dt, hi = h0 + dh, wi = w0 + dw;
                if (ti >= 0 && ti < p.cv_Tin && hi >= 0 && hi < hb && wi >= 0 && wi < wb) ok |= 1u << tap;
              }
          cv_ok[h][t] = ok;
          const int hq = p.cv_up2x ? (h0 >> 1) : h0, wq = p.cv_up2x ? (w0 >> 1) : w0;
          cv_par[h][t] = p.cv_up2x ? (unsigned)((h0 & 1) | ((w0 & 1) << 1)) : 0u;
          arow[h][t] = ((bb * p.cv_Tin + t0) * p.cv_Hin + hq) * p.cv_Win + wq;
        }
        g = n0 + (srow >> 5) * 64 + h * 32 + (srow & 31);
        wrow[h][t] = (int)((g > p.N - 1 ? p.N - 1 : g) - n0);
      }
    }
  };
  int vid = p.vid_begin + (SPLIT ? (int)(blockIdx.x % (unsigned)p.vid_count) : (int)blockIdx.x);
  setup_tile(vid);
  auto stage_a = [&](auto mq_c, int kts, int buf) {
    constexpr int mq = decltype(mq_c)::value;
    if constexpr (SPLIT) kts += k_off;
    const bool lora = kts >= p.nk1;  // the rank-r pair (a2, w2) supplies the last nk2 K tiles
    const char* base = lora ? (const char*)p.a2 + a2_tile + (int64_t)(kts - p.nk1) * 128 : (const char*)p.a + a_tile + (int64_t)kts * 128;
    const unsigned ldb = (unsigned)(lora ? p.lda2 : p.lda) * 2u;
    unsigned char* dst = smem + buf * BUF_BYTES + mq * SLOT_BYTES + wave * 2048;
    if constexpr (CONV) {
      const int tap = cvc_tap[mq], dh = cvc_dh[mq], dw = cvc_dw[mq], chunk = cvc_chunk[mq];
      const int dpix_t = cvc_dt[mq] * p.cv_Hin * p.cv_Win;
      if (++cvc_chunk[mq] == p.cv_cpt) {            // advance to the K tile this stream stages next
        cvc_chunk[mq] = 0; ++cvc_tap[mq];
        if (++cvc_dw[mq] == p.cv_kw) { cvc_dw[mq] = 0; if (++cvc_dh[mq] == p.cv_kh) { cvc_dh[mq] = 0; ++cvc_dt[mq]; } }
      }
      const uint64_t abase = (uint64_t)(p.a + chunk * 64), zbase = (uint64_t)p.cv_zero;
      const unsigned row_b = (unsigned)p.lda * 2u;
      const int dpix = dpix_t + dh * p.cv_Win + dw;
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        int pix;
        if (p.cv_up2x) {
          const int ih = (int)((cv_par[mq][t] & 1u) + (unsigned)dh) >> 1, iw = (int)((cv_par[mq][t] >> 1) + (unsigned)dw) >> 1;
          pix = arow[mq][t] + dpix_t + __mul24(ih, p.cv_Win) + iw;
        } else {
          pix = arow[mq][t] + dpix;
        }
        const bool ok = (cv_ok[mq][t] >> tap) & 1u;
        const uint64_t src = (ok ? abase : zbase) + (uint64_t)(ok ? (unsigned)pix : 0u) * row_b + swz[t];
        __builtin_amdgcn_global_load_lds((gbl_void*)src, (lds_void*)(dst + t * 1024), 16, 0, 0);
      }
      return;
    }
#pragma unroll
    for (int t = 0; t < 2; ++t)
      __builtin_amdgcn_global_load_lds((gbl_void*)(base + ((unsigned)arow[mq][t] * ldb + swz[t])), (lds_void*)(dst + t * 1024),
                                       16, 0, 0);
  };
  auto stage_w = [&](auto nq_c, int kts, int buf) {
    constexpr int nq = decltype(nq_c)::value;
    if constexpr (SPLIT) kts += k_off;
    const bool lora = kts >= p.nk1;
    const char* base = lora ? (const char*)p.w2 + w2_tile + (int64_t)(kts - p.nk1) * 128 : (const char*)p.w + w_tile + (int64_t)kts * 128;
    const unsigned ldb = (unsigned)(lora ? p.ldw2 : p.ldw) * 2u;
    unsigned char* dst = smem + buf * BUF_BYTES + (2 + nq) * SLOT_BYTES + wave * 2048;
#pragma unroll
    for (int t = 0; t < 2; ++t)
      __builtin_amdgcn_global_load_lds((gbl_void*)(base + ((unsigned)wrow[nq][t] * ldb + swz[t])), (lds_void*)(dst + t * 1024),
                                       16, 0, 0);
  };

  // ---- fragment read addresses: slot row (wr * 64 + 16 i + r16) of an A slot, (wc * 32 + 16 j + r16) of a W slot ----
  const int sw = (r16 >> 1) & 7;
  int a_rd[2], w_rd[2];  // [ks]
#pragma unroll
  for (int ks = 0; ks < 2; ++ks) {
    a_rd[ks] = (wr * 64 + r16) * 128 + ((4 * ks + q) ^ sw) * 16;
    w_rd[ks] = 2 * SLOT_BYTES + (wc * 32 + r16) * 128 + ((4 * ks + q) ^ sw) * 16;
  }

  f32x4v acc[8][4];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4v{0.f, 0.f, 0.f, 0.f};
  bf16x8 af[4][2];     // [i][ks]      activations of the current 64-row half (B operand)
  bf16x8 wf[2][2][2];  // [nq][j][ks]  weights of both 32-column halves (A operand)

  using C0 = std::integral_constant<int, 0>;
  using C1 = std::integral_constant<int, 1>;
  // one phase.  P = 1..4, STAGE = issue this phase's half-tile, VM = vmcnt to wait for (-1: none); buf = buffer of K tile kt
  // (wave-uniform; the fragment read bases a_rd / w_rd already point into it)
  auto phase = [&](auto P_c, auto STAGE_c, auto VM_c, int kt, int buf) {
    constexpr int P = decltype(P_c)::value, VM = decltype(VM_c)::value;
    constexpr bool STAGE = decltype(STAGE_c)::value != 0;
    if constexpr (P == 1 || P == 2) {
      constexpr int nq = P - 1;
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
          wf[nq][j][ks] = *reinterpret_cast<const bf16x8*>(smem + w_rd[ks] + nq * SLOT_BYTES + j * 2048);
    }
    if constexpr (P == 1 || P == 3) {
      constexpr int mq = P == 1 ? 0 : 1;
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
          af[i][ks] = *reinterpret_cast<const bf16x8*>(smem + a_rd[ks] + mq * SLOT_BYTES + i * 2048);
    }
    if constexpr (STAGE) {
      if constexpr (P == 1) stage_w(C1{}, kt + 1, buf ^ 1);
      if constexpr (P == 2) stage_a(C1{}, kt + 1, buf ^ 1);
      if constexpr (P == 3) stage_a(C0{}, kt + 2, buf);
      if constexpr (P == 4) stage_w(C0{}, kt + 2, buf);
    }
    if constexpr (VM >= 0) wait_vmcnt<VM>();
    GSTAMP();
    __builtin_amdgcn_s_barrier();
    GSTAMP();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    GSTAMP();
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_setprio(1);
    constexpr int mq = (P >= 3) ? 1 : 0;
    constexpr int nq = (P == 2 || P == 3) ? 1 : 0;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
          acc[4 * mq + i][2 * nq + j] =
              __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[nq][j][ks], af[i][ks], acc[4 * mq + i][2 * nq + j], 0, 0, 0);
    __builtin_amdgcn_s_setprio(0);
    __builtin_amdgcn_sched_barrier(0);
    GSTAMP();
    __builtin_amdgcn_s_barrier();
    GSTAMP();
  };
  using V8 = std::integral_constant<int, 8>;
  using V6 = std::integral_constant<int, 6>;
  using V4 = std::integral_constant<int, 4>;
  using V2 = std::integral_constant<int, 2>;
  using V0 = std::integral_constant<int, 0>;
  using VN = std::integral_constant<int, -1>;
  using P1 = std::integral_constant<int, 1>;
  using P2 = std::integral_constant<int, 2>;
  using P3 = std::integral_constant<int, 3>;
  using P4 = std::integral_constant<int, 4>;
  auto flip = [&]() {  // fragment read bases -> the other buffer
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) { a_rd[ks] ^= BUF_BYTES; w_rd[ks] ^= BUF_BYTES; }
  };

  // ---- prologue: the steady-state issue order A-mq0, W-nq0, W-nq1, A-mq1 of tile 0, then A-mq0, W-nq0 of tile 1 ----
  auto prologue = [&]() {
    if constexpr (CONV) {
#pragma unroll
      for (int m = 0; m < 2; ++m) { cvc_chunk[m] = 0; cvc_dw[m] = 0; cvc_dh[m] = 0; cvc_dt[m] = 0; cvc_tap[m] = 0; }
    }
    stage_a(C0{}, 0, 0);
    stage_w(C0{}, 0, 0);
    stage_w(C1{}, 0, 0);
    stage_a(C1{}, 0, 0);
    stage_a(C0{}, 1, 1);
    stage_w(C0{}, 1, 1);
  };
  prologue();
  for (;;) {
    // A-mq0 and W-nq0 of K tile 0 have landed (persistent: the previous tile's stores are younger than the prologue, so
    // this also retires the whole prologue and all but 8 of those stores - conservative, never early)
    wait_vmcnt<8>();
    __builtin_amdgcn_s_barrier();
    if (wr == 1) __builtin_amdgcn_s_barrier();  // the lower wave row runs one barrier behind

    int kt = 0;
    for (; kt < nk - 2; ++kt) {
      const int buf = kt & 1;
      phase(P1{}, C1{}, V8{}, kt, buf);
      phase(P2{}, C1{}, V8{}, kt, buf);
      phase(P3{}, C1{}, V8{}, kt, buf);
      phase(P4{}, C1{}, V8{}, kt, buf);
      flip();
    }
    {  // K tile nk-2: nothing left to stage for tile nk
      const int buf = kt & 1;
      phase(P1{}, C1{}, V8{}, kt, buf);
      phase(P2{}, C1{}, V8{}, kt, buf);
      phase(P3{}, C0{}, V6{}, kt, buf);
      phase(P4{}, C0{}, V4{}, kt, buf);
      flip();
      ++kt;
    }
    {  // K tile nk-1
      const int buf = kt & 1;
      phase(P1{}, C0{}, V2{}, kt, buf);
      phase(P2{}, C0{}, V0{}, kt, buf);
      phase(P3{}, C0{}, VN{}, kt, buf);
      phase(P4{}, C0{}, VN{}, kt, buf);
    }
    if (wr == 0) __builtin_amdgcn_s_barrier();  // balance the stagger

    const int64_t mw = m0 + wr * 128, nw = n0 + wc * 64;
    if constexpr (SPLIT) {
      // partial sums of this K slice, in register order: 32 coalesced 16-byte stores per lane
      float* wsp = p.ws + ((int64_t)blockIdx.x * 32 * 512 + tid) * 4;
#pragma unroll
      for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) *reinterpret_cast<f32x4v*>(wsp + (int64_t)(i * 4 + j) * 512 * 4) = acc[i][j];
      break;
    } else if constexpr (PERSIST) {
      // every wave is past its last LDS read: both buffers are free, so the next tile's pipeline fills under this epilogue
      const int next = vid + (int)gridDim.x;
      const bool more = next < nwg;
      if (more) {
        if (kt & 1) flip();  // fragment read bases back to buffer 0
        setup_tile(next);
        prologue();
      }
      if (fast_epi && g4_fast_epilogue_ok<EPI, 4>(p, mw, nw)) g4_fast_epilogue<EPI, 4>(p, acc, mw, nw, r16, q);
      else gemm16_epilogue<8, 4, EPI>(p, acc, mw, nw, r16, q);
      if (!more) break;
      vid = next;
#pragma unroll
      for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4v{0.f, 0.f, 0.f, 0.f};
    } else {
      if (fast_epi && g4_fast_epilogue_ok<EPI, 4>(p, mw, nw)) g4_fast_epilogue<EPI, 4>(p, acc, mw, nw, r16, q);
      else gemm16_epilogue<8, 4, EPI>(p, acc, mw, nw, r16, q);
      break;
    }
  }
}

// Adds the K slices of the tail tiles (same thread -> accumulator mapping as gemm8p_nt_kernel) and runs the normal epilogue.
template <int EPI>
__global__ __launch_bounds__(512) void gemm8p_splitk_reduce_kernel(const GemmParams p) {
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wr = wave >> 2, wc = wave & 3;
  const int r16 = lane & 15, q = lane >> 4;
  int tm, tn;
  gemm_tile_coords(p, p.vid_begin + (int)blockIdx.x, tm, tn);
  f32x4v acc[8][4];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4v{0.f, 0.f, 0.f, 0.f};
  for (int sl = 0; sl < p.splitk; ++sl) {
    const float* wsp = p.ws + (((int64_t)sl * p.vid_count + blockIdx.x) * 32 * 512 + tid) * 4;
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[i][j] += *reinterpret_cast<const f32x4v*>(wsp + (int64_t)(i * 4 + j) * 512 * 4);
  }
  gemm16_epilogue<8, 4, EPI>(p, acc, (int64_t)tm * 256 + wr * 128, (int64_t)tn * 256 + wc * 64, r16, q);
}

// caller-supplied fp32 workspace for the split-K tail (lcv_gemm_set_workspace); one per process, used on one stream at a time
static float* g_gemm_ws = nullptr;
static int64_t g_gemm_ws_bytes = 0;

extern "C" int lcv_gemm_set_workspace(void* ws, int64_t bytes) {
  LCV_CHECK_ARG((ws == nullptr) == (bytes == 0) && bytes >= 0 && ((uintptr_t)ws % 16) == 0, "gemm_set_workspace: bad arguments");
  g_gemm_ws = (float*)ws;
  g_gemm_ws_bytes = bytes;
  return LCV_OK;
}

// Tail of a persistent launch: ntiles = r * 256 + t tiles on 256 CUs cost r + 1 tile-times although the last round keeps
// only t CUs busy.  With a workspace a SMALL tail is split s ways along K into t*s work items (one partial round of ~1/s
// tile-time), summed and finished by the reduce kernel.  Returns the chosen s (1 = leave the tail alone).
// Cost model in tile-times, fitted to in-process A/B runs (scratch/bench_kernels.py gemm_tail): the two extra kernel
// boundaries cost ~0.15, every work item moves 2 x 256 KiB of fp32 partials through HBM (~0.0013 each), a slice costs its K
// tiles + 2 of pipeline fill.  Measured: t = 16 (784 tiles, the 480p generation shapes) +4 % / +6 %; t = 144 (400 tiles)
// -23 % when split 5 ways - the partial sums of a fat tail cost more than its idle CUs, so only thin tails qualify.
static int choose_tail_split(int t, int nk, int64_t ws_bytes) {
  if (t <= 0 || t > 32) return 1;
  const char* e = lcv_knob("LCV_GEMM_SPLITK_TAIL");
  if (e && e[0] == '0') return 1;
  int best_s = 1;
  double best = 1.0;
  for (int s = 2; s <= 16; ++s) {
    const int nks = (nk + s - 1) / s;
    if (nks < 4 || nk - (s - 1) * nks < 2) break;        // slices of >= 4 K tiles (the pipeline needs 2 to fill), none empty
    if ((int64_t)t * s * 512 * 128 * 4 > ws_bytes) break;
    const double cost = (double)((t * s + 255) / 256) * (nks + 2) / (double)nk + 0.15 + 0.0013 * t * s;
    if (cost < best - 0.1) { best = cost; best_s = s; }
  }
  return best_s;
}

template <int EPI, bool PERSIST>
static int launch_gemm8p(GemmParams& p, hipStream_t s) {
  p.tiles_m = (int)((p.M + 255) / 256);
  // tile rows per group of the tile order (A/B knob LCV_GEMM_GROUP_M): 6 measured best for the persistent kernel at the K3
  // projection shapes (4, 6, 8, 16 -> 1267, 1285, 1240, 1172 TF/s on the qkv GEMM, in one process)
  { const char* ge = lcv_knob("LCV_GEMM_GROUP_M"); p.group_m = ge ? atoi(ge) : 6; if (p.group_m < 1) p.group_m = 6; }
  p.tiles_n = (int)((p.N + 255) / 256);
  const size_t lds = 2 * 65536;
  auto kern = gemm8p_nt_kernel<EPI, PERSIST>;
  // (function-local static: initialised once, thread-safe)
  static const bool attr_ok = !(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess);
  if (!attr_ok) {
      lcv_set_error("gemm_nt: cannot raise dynamic LDS to %zu", lds);
      return LCV_EDEVICE;
  }
  const int ntiles = p.tiles_m * p.tiles_n;
  p.vid_begin = 0; p.vid_count = ntiles; p.splitk = 1; p.nk_split = 0; p.ws = nullptr;
  { const char* fe = lcv_knob("LCV_GEMM_FAST_EPI"); p.fast_epi = (fe && fe[0] == '0') ? 0 : 1; }
  if (PERSIST && ntiles > 256 && g_gemm_ws != nullptr) {
    const int t = ntiles % 256, nk = p.nk1 + p.nk2;
    const int sk = choose_tail_split(t, nk, g_gemm_ws_bytes);
    if (sk > 1) {
      auto skern = gemm8p_nt_kernel<LCV_EPI_NONE, false, true>;
      // (function-local static: initialised once, thread-safe)
      static const bool sattr_ok = !(hipFuncSetAttribute((const void*)skern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess);
      if (!sattr_ok) {
          lcv_set_error("gemm_nt: cannot raise dynamic LDS to %zu", lds);
          return LCV_EDEVICE;
      }
      p.vid_count = ntiles - t;                           // full rounds: every CU busy to the end
      hipLaunchKernelGGL(kern, dim3(256), dim3(512), lds, s, p);
      LCV_LAUNCH_CHECK("gemm8p_nt");
      p.vid_begin = ntiles - t; p.vid_count = t; p.splitk = sk; p.nk_split = (nk + sk - 1) / sk; p.ws = g_gemm_ws;
      hipLaunchKernelGGL(skern, dim3((unsigned)(t * sk)), dim3(512), lds, s, p);
      LCV_LAUNCH_CHECK("gemm8p_nt_splitk");
      hipLaunchKernelGGL(gemm8p_splitk_reduce_kernel<EPI>, dim3((unsigned)t), dim3(512), 0, s, p);
      LCV_LAUNCH_CHECK("gemm8p_splitk_reduce");
      return LCV_OK;
    }
  }
  unsigned grid = (unsigned)ntiles;
  if (PERSIST && grid > 256) grid = 256;  // one workgroup per CU (the LDS image allows no more); a multiple of 8 XCDs
  hipLaunchKernelGGL(kern, dim3(grid), dim3(512), lds, s, p);
  LCV_LAUNCH_CHECK("gemm8p_nt");
  return LCV_OK;
}

// The convolution form of the 8-phase kernel (N >= 192 stages of the VAE): persistent, no split-K tail.
template <int EPI>
static int launch_conv8p(GemmParams& p, hipStream_t s) {
  p.tiles_m = (int)((p.M + 255) / 256);
  p.group_m = 6;
  p.tiles_n = (int)((p.N + 255) / 256);
  const size_t lds = 2 * 65536;
  auto kern = gemm8p_nt_kernel<EPI, true, false, true>;
  // (function-local static: initialised once, thread-safe)
  static const bool attr_ok = !(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess);
  if (!attr_ok) {
      lcv_set_error("conv3d: cannot raise dynamic LDS to %zu", lds);
      return LCV_EDEVICE;
  }
  const int ntiles = p.tiles_m * p.tiles_n;
  p.vid_begin = 0; p.vid_count = ntiles; p.splitk = 1; p.nk_split = 0; p.ws = nullptr;
  unsigned grid = (unsigned)ntiles;
  if (grid > 256) grid = 256;
  hipLaunchKernelGGL(kern, dim3(grid), dim3(512), lds, s, p);
  LCV_LAUNCH_CHECK("conv8p_igemm");
  return LCV_OK;
}

#include "gemm4k.h"

// four waves x 128 x 128 on 64-deep K tiles in 128-byte rows (gemm4k.h, round 4), persistent over all tiles of the launch
template <int EPI>
static int launch_gemm4k(GemmParams& p, hipStream_t s) {
  p.tiles_m = (int)((p.M + 255) / 256);
  // tile rows per group of the XCD-contiguous tile order: 3 measured best for this kernel (1 ... 64 swept in one process at the
  // qkv / w13 / proj shapes, profiles/r04_gemm_ab.md: 1435 / 1358 / 1411 TF/s at 3, 1398 / 1353 / 1420 at 6, 1254 / 1251 / 1275 at 16)
  { const char* ge = lcv_knob("LCV_GEMM_GROUP_M"); p.group_m = ge ? atoi(ge) : 3; if (p.group_m < 1) p.group_m = 3; }
  p.tiles_n = (int)((p.N + 255) / 256);
  const size_t lds = 2 * 65536;
  auto kern = gemm4k_nt_kernel<EPI>;
  // (function-local static: initialised once, thread-safe)
  static const bool attr_ok = !(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess);
  if (!attr_ok) {
      lcv_set_error("gemm_nt: cannot raise dynamic LDS to %zu", lds);
      return LCV_EDEVICE;
  }
  const int ntiles = p.tiles_m * p.tiles_n;
  p.vid_begin = 0; p.vid_count = ntiles; p.splitk = 1; p.nk_split = 0; p.ws = nullptr;
  hipLaunchKernelGGL(kern, dim3((unsigned)(ntiles > 256 ? 256 : ntiles)), dim3(256), lds, s, p);
  LCV_LAUNCH_CHECK("gemm4k_nt");
  return LCV_OK;
}

template <int BM, int BN, int WR, int WC, int EPI, bool CONV, int NST = 2>
static int launch_gemm16(GemmParams& p, hipStream_t s) {
  using Cfg = GemmCfg<BM, BN, WR, WC>;
  p.tiles_m = (int)((p.M + BM - 1) / BM);
  p.group_m = 8;
  p.tiles_n = (int)((p.N + BN - 1) / BN);
  const size_t lds = NST * Cfg::STAGE_BYTES;
  static_assert(NST * Cfg::STAGE_BYTES <= 163840, "gemm16: LDS image");
  auto kern = gemm16_nt_kernel<BM, BN, WR, WC, EPI, CONV, NST>;
  // (function-local static: initialised once, thread-safe)
  static const bool attr_ok = !(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess);
  if (!attr_ok) {
      lcv_set_error("gemm_nt: cannot raise dynamic LDS to %zu", lds);
      return LCV_EDEVICE;
  }
  hipLaunchKernelGGL(kern, dim3((unsigned)(p.tiles_m * p.tiles_n)), dim3(Cfg::NT), lds, s, p);
  LCV_LAUNCH_CHECK(CONV ? "conv16_igemm" : "gemm16_nt");
  return LCV_OK;
}

template <int BM, int BN, int WR, int WC, int EPI, bool CONV>
static int launch_gemm(GemmParams& p, hipStream_t s) {
  using Cfg = GemmCfg<BM, BN, WR, WC>;
  p.tiles_m = (int)((p.M + BM - 1) / BM);
  p.group_m = 8;
  p.tiles_n = (int)((p.N + BN - 1) / BN);
  const size_t lds = 2 * Cfg::STAGE_BYTES;
  auto kern = gemm_nt_kernel<BM, BN, WR, WC, EPI, CONV>;
  // (function-local static: initialised once, thread-safe)
  static const bool attr_ok = !(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess);
  if (!attr_ok) {
      lcv_set_error("gemm_nt: cannot raise dynamic LDS to %zu", lds);
      return LCV_EDEVICE;
  }
  hipLaunchKernelGGL(kern, dim3((unsigned)(p.tiles_m * p.tiles_n)), dim3(Cfg::NT), lds, s, p);
  LCV_LAUNCH_CHECK(CONV ? "conv_igemm" : "gemm_nt");
  return LCV_OK;
}

template <int EPI>
static int dispatch_tile(GemmParams& p, hipStream_t s) {
  // v_mfma_f32_16x16x32_bf16 kernels by default: 256x256 tiles for the big token-side projections (measured
  // 1062-1118 TF/s at K3 shapes vs 982-1004 for the 32x32x16 kernel), 128x128 for small M or N.
  // LCV_GEMM_TILE = 6 | 7 forces 256 / 128 (16x16x32); 2 | 1 the same tiles on the 32x32x16 kernel (A/B runs, tests).
  const char* force = lcv_knob("LCV_GEMM_TILE");
  int mode = (p.M >= 2048 && p.N >= 1024) ? 6 : 7;
  // 8-phase ping-pong schedule on the same tile; its LDS-DMA sources are 32-bit byte offsets from the operand base
  const bool ok8 = p.nk1 >= 2 && (uint64_t)p.M * p.lda * 2 < (1ull << 32) && (uint64_t)p.N * p.ldw * 2 < (1ull << 32) &&
                   (p.nk2 == 0 || ((uint64_t)p.M * p.lda2 * 2 < (1ull << 32) && (uint64_t)p.N * p.ldw2 * 2 < (1ull << 32)));
  // round 3: the default 8-phase kernel addresses rows RELATIVE to its tile (64-bit tile base in SGPRs), so only a 256-row
  // panel has to fit 32 bits: any M (M = 352 800 x K = 11 008 at the literal 49x90x160 config used to fall back to gemm16)
  const bool ok8t = p.nk1 >= 2 && (uint64_t)256 * p.lda * 2 < (1ull << 31) && (uint64_t)256 * p.ldw * 2 < (1ull << 31) &&
                    (p.nk2 == 0 || ((uint64_t)256 * p.lda2 * 2 < (1ull << 31) && (uint64_t)256 * p.ldw2 * 2 < (1ull << 31)));
  if (mode == 6 && ok8t) mode = 9;  // persistent workgroups (identical to 8 when there are no more tiles than CUs)
  // round 4: the four-wave K64 kernel (gemm4k.h) where its single epilogue serves the call (bf16 output, N a multiple of 256,
  // 16-byte aligned operands of the epilogue): +4 ... +20 % over the 8-phase kernel at every DiT shape, bit-identical
  // (profiles/r04_gemm_ab.md); LCV_GEMM_TILE=9 selects the 8-phase kernel again
  if (mode == 9 && ok8) mode = 10;
  if (force) mode = force[0] == 'k' ? 10 : force[0] - '0';
  if constexpr (EPI == LCV_EPI_NONE || EPI == LCV_EPI_GATE_RESIDUAL || EPI == LCV_EPI_SWIGLU) {
    if (mode == 10 && ok8 && gemm4k_eligible<EPI>(p)) return launch_gemm4k<EPI>(p, s);
  }
  if (mode == 10) mode = ok8t ? 9 : 6;
  if (mode == 8 && ok8t) return launch_gemm8p<EPI, false>(p, s);
  if (mode == 9 && ok8t) return launch_gemm8p<EPI, true>(p, s);
  if (mode == 6 || mode == 8 || mode == 9) return launch_gemm16<256, 256, 2, 4, EPI, false>(p, s);
  if (mode == 7) return launch_gemm16<128, 128, 2, 2, EPI, false>(p, s);
  if (mode == 2) return launch_gemm<256, 256, 2, 4, EPI, false>(p, s);
  return launch_gemm<128, 128, 2, 2, EPI, false>(p, s);
}

#include "conv_wide.h"

template <int EPI>
static int dispatch_conv(GemmParams& p, hipStream_t s) {
  // wide stages: the 8-phase kernel when its pipeline has K tiles to fill (>= 2)
  // wide stages.  LCV_CONV_8P=1: the 8-phase kernel with the gather in its A stream - bit-identical to the two-stage kernel and
  // no faster on the VAE's shapes (802 vs 816 TF/s on 192 -> 192 at 360 x 640), so it stays opt-in.
  const char* e8 = lcv_knob("LCV_CONV_8P");
  if (p.N >= 192 && p.nk1 >= 2 && e8 && e8[0] == '1') return launch_conv8p<EPI>(p, s);
  // Cout = 192 / 384: 192-column tiles instead of 256-column tiles of which a quarter multiplies padding
  const char* e192 = lcv_knob("LCV_CONV_N192");
  if (p.N >= 192 && p.N % 192 == 0 && !(e192 && e192[0] == '0')) {
    // LCV_CONV_N192=3: 192 x 192 tiles on a ring of three buffers with a counted wait - bit-identical and SLOWER (873 vs 919,
    // 952 vs 1011 TF/s): these kernels do not wait for the round trip of their requests
    if (e192 && e192[0] == '3') return launch_gemm16<192, 192, 2, 4, EPI, true, 3>(p, s);
    if (e192 && e192[0] == '2') return launch_gemm16<256, 192, 2, 4, EPI, true>(p, s);      // every wave stages and multiplies
    return launch_conv_wide<EPI>(p, s);                                                     // loader waves + MFMA waves
  }
  if (p.N >= 192) return launch_gemm16<256, 256, 2, 4, EPI, true>(p, s);
  return launch_gemm16<128, 128, 2, 2, EPI, true>(p, s);
}

extern "C" int lcv_gemm_nt(const void* a, const void* w, const void* bias, const void* a2, const void* w2,
                           void* c, int64_t M, int64_t N, int64_t K, int64_t K2, int64_t lda, int64_t ldw,
                           int64_t lda2, int64_t ldw2, int64_t ldc, int epilogue, int out_f32,
                           const void* resid, const float* mod, int64_t rows_per_frame, int64_t mod_stride,
                           int64_t gate_off, void* stream) {
  LCV_CHECK_ARG(a && w && c, "gemm_nt: null pointer");
  LCV_CHECK_ARG(M >= 0 && N > 0 && K > 0 && K % 64 == 0, "gemm_nt: K=%ld must be a positive multiple of 64", (long)K);
  LCV_CHECK_ARG(K2 >= 0 && K2 % 64 == 0, "gemm_nt: K2=%ld must be a multiple of 64", (long)K2);
  LCV_CHECK_ARG(K2 == 0 || (a2 && w2), "gemm_nt: K2 > 0 needs a2 and w2");
  LCV_CHECK_ARG(lda % 8 == 0 && ldw % 8 == 0 && (K2 == 0 || (lda2 % 8 == 0 && ldw2 % 8 == 0)),
                "gemm_nt: operand row strides must be multiples of 8 elements (16 bytes)");
  LCV_CHECK_ARG(((uintptr_t)a % 16 == 0) && ((uintptr_t)w % 16 == 0), "gemm_nt: operands must be 16-byte aligned");
  if (M == 0) return LCV_OK;
  GemmParams p{};
  p.a = (const bf16_t*)a; p.w = (const bf16_t*)w; p.bias = (const bf16_t*)bias;
  p.a2 = (const bf16_t*)a2; p.w2 = (const bf16_t*)w2; p.c = c;
  p.M = M; p.N = N; p.nk1 = (int)(K / 64); p.nk2 = (int)(K2 / 64);
  p.lda = lda; p.ldw = ldw; p.lda2 = lda2; p.ldw2 = ldw2; p.ldc = ldc; p.out_f32 = out_f32;
  p.resid = (const bf16_t*)resid; p.gate = mod ? mod + gate_off : nullptr;
  p.rows_per_frame = rows_per_frame > 0 ? rows_per_frame : 1; p.mod_stride = mod_stride;
  p.cv_zero = nullptr; p.cv_up2x = 0; p.cv_cpt = 1;
  p.cv_st = p.cv_sh = p.cv_sw = 1; p.cv_pt = p.cv_ph = p.cv_pw = 0;
  hipStream_t s = (hipStream_t)stream;
  switch (epilogue) {
    case LCV_EPI_NONE: return dispatch_tile<LCV_EPI_NONE>(p, s);
    case LCV_EPI_SWIGLU:
      LCV_CHECK_ARG(N % 64 == 0 && !out_f32, "gemm_nt: SwiGLU epilogue needs N %% 64 == 0 and bf16 output");
      return dispatch_tile<LCV_EPI_SWIGLU>(p, s);
    case LCV_EPI_GATE_RESIDUAL:
      LCV_CHECK_ARG(resid != nullptr, "gemm_nt: gate-residual epilogue needs resid");
      return dispatch_tile<LCV_EPI_GATE_RESIDUAL>(p, s);
    case LCV_EPI_GELU_TANH: return dispatch_tile<LCV_EPI_GELU_TANH>(p, s);
    case LCV_EPI_SILU: return dispatch_tile<LCV_EPI_SILU>(p, s);
    default:
      lcv_set_error("gemm_nt: unknown epilogue %d", epilogue);
      return LCV_EINVAL;
  }
}

static thread_local const char* g_last_conv_kernel = "none";
extern "C" const char* lcv_conv3d_last_kernel(void) { return g_last_conv_kernel; }

#include "conv_rows.h"

// ---------------------------------------------------------------------------
// Causal 3-D convolution as an implicit GEMM on the same MFMA core (VAE decoder).
//   x   [B, Tin, Hin, Win, ldx]  channels-last bf16, Cin % 32 == 0 valid channels, ldx = roundup64(Cin), pad channels zero
//   w   [Cout, kt*kh*kw*ldx]     K ordered (dt, dh, dw, cin)
// Cin % 96 == 0 with Cout <= 96 at stride 1 takes the row-tile kernel (conv_rows.h), everything else the implicit GEMM.
//   out [B, T, H, W, ldc] with T = Tin, (H, W) = (Hin, Win) or doubled when up2x (nearest upsample fused in the gather)
// Temporal padding is causal (kt-1 zero frames in front), spatial padding kh/2, kw/2 zeros.
// resid (nullable, same layout as out): out = resid + bf16(conv + bias)  (residual-block tail).
// ---------------------------------------------------------------------------
static int conv3d_impl(const void* x, const void* w, const void* bias, const void* resid, void* out, const void* zero_page,
                       int64_t B, int64_t Tin, int64_t Hin, int64_t Win, int64_t Cin, int64_t Cout, int64_t ldc, int kt, int kh,
                       int kw, int up2x, int st, int sh, int sw, int pt, int ph, int pw, int64_t Tout, int64_t Hout,
                       int64_t Wout, void* stream) {
  // the kernel keeps one validity bit per tap in a 32-bit word and 32-bit pixel indices per staged row
  LCV_CHECK_ARG(kt * kh * kw <= 32, "conv3d: %d taps, at most 32 supported", kt * kh * kw);
  LCV_CHECK_ARG(B * Tin * Hin * Win < (int64_t(1) << 31) && Cin <= 4096 && Cout <= 65536 && Win * ((Cin + 63) / 64 * 64) * 2 < (int64_t(1) << 30) &&
                    Cout * (int64_t)kt * kh * kw * ((Cin + 63) / 64 * 64) * 2 < (int64_t(1) << 31),
                "conv3d: input of %ld pixels is too large", (long)(B * Tin * Hin * Win));
  GemmParams p{};
  p.a = (const bf16_t*)x; p.w = (const bf16_t*)w; p.bias = (const bf16_t*)bias; p.a2 = nullptr; p.w2 = nullptr;
  p.c = out;
  p.cv_T = (int)Tout; p.cv_H = (int)Hout; p.cv_W = (int)Wout;
  p.cv_Tin = (int)Tin; p.cv_Hin = (int)Hin; p.cv_Win = (int)Win;
  const int64_t ldx = (Cin + 63) / 64 * 64;   // pixel stride: channels are padded to a multiple of 64 (pads zero, also in w)
  p.cv_kt = kt; p.cv_kh = kh; p.cv_kw = kw; p.cv_cpt = (int)(ldx / 64); p.cv_up2x = up2x;
  p.cv_st = st; p.cv_sh = sh; p.cv_sw = sw; p.cv_pt = pt; p.cv_ph = ph; p.cv_pw = pw;
  p.cv_zero = (const bf16_t*)zero_page;
  p.M = B * p.cv_T * (int64_t)p.cv_H * p.cv_W; p.N = Cout;
  p.nk1 = kt * kh * kw * p.cv_cpt; p.nk2 = 0;
  p.lda = ldx; p.ldw = (int64_t)kt * kh * kw * ldx; p.lda2 = 0; p.ldw2 = 0; p.ldc = ldc; p.out_f32 = 0;
  p.resid = (const bf16_t*)resid; p.gate = nullptr; p.rows_per_frame = 1; p.mod_stride = 0;
  if (p.M == 0) return LCV_OK;
  hipStream_t s = (hipStream_t)stream;
  if (conv_rows_applies(p, Cin)) {
    p.cv_cpt = (int)(Cin / 96);
    { const char* eo = lcv_knob("LCV_CONV_ROWS_ORDER"); p.splitk = (eo && eo[0] == 'w') ? 1 : 0; }   // tile sequence (conv_rows.h)
    g_last_conv_kernel = p.N <= 16 ? "conv_rows<256x16>" : "conv_rows<256x96>";
    if (p.N <= 16) return resid ? launch_conv_rows<8, 1, 2, 1, LCV_EPI_GATE_RESIDUAL>(p, s) : launch_conv_rows<8, 1, 2, 1, LCV_EPI_NONE>(p, s);
    return resid ? launch_conv_rows<4, 2, 4, 3, LCV_EPI_GATE_RESIDUAL>(p, s) : launch_conv_rows<4, 2, 4, 3, LCV_EPI_NONE>(p, s);
  }
  { const char* e8 = lcv_knob("LCV_CONV_8P"); const char* e192 = lcv_knob("LCV_CONV_N192");
    g_last_conv_kernel = p.N < 192 ? "conv16_igemm<128x128>"
                         : (p.nk1 >= 2 && e8 && e8[0] == '1') ? "conv8p_igemm<256x256>"
                         : (p.N % 192 == 0 && !(e192 && e192[0] == '0')) ? ((e192 && e192[0] == '3') ? "conv16_igemm<192x192x3>" : (e192 && e192[0] == '2') ? "conv16_igemm<256x192>" : "conv_wide<256x192>")
                                                                         : "conv16_igemm<256x256>"; }
  if (resid) return dispatch_conv<LCV_EPI_GATE_RESIDUAL>(p, s);
  return dispatch_conv<LCV_EPI_NONE>(p, s);
}

extern "C" int lcv_causal_conv3d(const void* x, const void* w, const void* bias, const void* resid, void* out,
                                 const void* zero_page, int64_t B, int64_t Tin, int64_t Hin, int64_t Win,
                                 int64_t Cin, int64_t Cout, int64_t ldc, int kt, int kh, int kw, int up2x,
                                 void* stream) {
  LCV_CHECK_ARG(x && w && out && zero_page, "causal_conv3d: null pointer");
  LCV_CHECK_ARG(Cin > 0 && Cin % 32 == 0, "causal_conv3d: Cin=%ld must be a multiple of 32", (long)Cin);
  LCV_CHECK_ARG(kt >= 1 && kh >= 1 && kw >= 1 && (kh & 1) && (kw & 1), "causal_conv3d: odd spatial kernels only");
  LCV_CHECK_ARG(ldc >= Cout, "causal_conv3d: ldc < Cout");
  return conv3d_impl(x, w, bias, resid, out, zero_page, B, Tin, Hin, Win, Cin, Cout, ldc, kt, kh, kw, up2x, 1, 1, 1, kt - 1,
                     kh >> 1, kw >> 1, Tin, up2x ? 2 * Hin : Hin, up2x ? 2 * Win : Win, stream);
}

// Strided form for the VAE encoder's downsampling convolutions: output pixel (t, h, w) reads input
// (t*st + dt, h*sh + dh, w*sw + dw) for dt < kt, dh < kh, dw < kw - NO front padding; taps past the input extent read
// zeros (that is the reference's ZeroPad2d((0,1,0,1)) + stride-2 3x3 conv, and its (3,1,1) stride-2 temporal conv over
// [last cached frame | chunk]).  The caller gives the output extent.
extern "C" int lcv_conv3d_strided(const void* x, const void* w, const void* bias, void* out, const void* zero_page,
                                  int64_t B, int64_t Tin, int64_t Hin, int64_t Win, int64_t Cin, int64_t Cout,
                                  int64_t ldc, int kt, int kh, int kw, int st, int sh, int sw, int64_t Tout,
                                  int64_t Hout, int64_t Wout, void* stream) {
  LCV_CHECK_ARG(x && w && out && zero_page, "conv3d_strided: null pointer");
  LCV_CHECK_ARG(Cin > 0 && Cin % 32 == 0, "conv3d_strided: Cin=%ld must be a multiple of 32", (long)Cin);
  LCV_CHECK_ARG(kt >= 1 && kh >= 1 && kw >= 1 && st >= 1 && sh >= 1 && sw >= 1, "conv3d_strided: bad kernel / stride");
  LCV_CHECK_ARG(ldc >= Cout, "conv3d_strided: ldc < Cout");
  LCV_CHECK_ARG(Tout >= 0 && Hout >= 0 && Wout >= 0 && (Tout - 1) * st < Tin && (Hout - 1) * sh < Hin && (Wout - 1) * sw < Win,
                "conv3d_strided: output extent reaches past the input");
  return conv3d_impl(x, w, bias, nullptr, out, zero_page, B, Tin, Hin, Win, Cin, Cout, ldc, kt, kh, kw, 0, st, sh, sw, 0, 0, 0,
                     Tout, Hout, Wout, stream);
}

// ---------------------------------------------------------------------------
// Small-M fp32 linear for the fp32 islands (t_embedder MLP, adaLN_modulation):
//   out[M,N] fp32 = act_in(a[M,K] fp32) @ W[N,K]^T (bf16 weights widened) + bias
// HBM-bound on the weight stream (adaLN: 24576 x 512 x 2 B = 25 MB per block);
// one wave per output column, activations staged once per workgroup in LDS.
// ---------------------------------------------------------------------------
template <int MAXM>
__global__ __launch_bounds__(256) void linear_f32_smallm_kernel(const float* __restrict__ a,
                                                                const bf16_t* __restrict__ w,
                                                                const bf16_t* __restrict__ bias,
                                                                float* __restrict__ out, int M, int64_t N,
                                                                int K, int act_in, int cols_per_wave) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  float* sa = reinterpret_cast<float*>(smem);  // [M][K]
  for (int i = threadIdx.x; i < M * K; i += 256) {
    float v = a[i];
    if (act_in == 1) v = silu_f(v);
    sa[i] = v;
  }
  __syncthreads();
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int64_t nbase = ((int64_t)blockIdx.x * 4 + wave) * cols_per_wave;
  for (int cc = 0; cc < cols_per_wave; ++cc) {
    const int64_t n = nbase + cc;
    if (n >= N) return;
    float acc[MAXM];
#pragma unroll
    for (int m = 0; m < MAXM; ++m) acc[m] = 0.f;
    for (int k = lane * 8; k < K; k += 512) {
      float wf[8];
      unpack8(*reinterpret_cast<const u16x8*>(w + n * K + k), wf);
#pragma unroll
      for (int m = 0; m < MAXM; ++m) {
        if (m < M) {
          const f32x4 x0 = *reinterpret_cast<const f32x4*>(sa + m * K + k);
          const f32x4 x1 = *reinterpret_cast<const f32x4*>(sa + m * K + k + 4);
          acc[m] += x0[0] * wf[0] + x0[1] * wf[1] + x0[2] * wf[2] + x0[3] * wf[3] + x1[0] * wf[4] +
                    x1[1] * wf[5] + x1[2] * wf[6] + x1[3] * wf[7];
        }
      }
    }
    const float b = bias ? bf2f(bias[n]) : 0.f;
#pragma unroll
    for (int m = 0; m < MAXM; ++m) {
      if (m < M) {
        const float s = wave_sum(acc[m]);
        if (lane == 0) out[(int64_t)m * N + n] = s + b;
      }
    }
  }
}

extern "C" int lcv_linear_f32_smallm(const float* a, const void* w, const void* bias, float* out, int64_t M,
                                     int64_t N, int64_t K, int act_in, void* stream) {
  LCV_CHECK_ARG(a && w && out, "linear_f32_smallm: null pointer");
  LCV_CHECK_ARG(K % 8 == 0 && K > 0 && N > 0, "linear_f32_smallm: K must be a positive multiple of 8");
  hipStream_t s = (hipStream_t)stream;
  // rows are processed in groups of <= 16 so the staged activations stay <= 64 KiB of LDS
  const int64_t max_rows = 16;
  LCV_CHECK_ARG(max_rows * K * 4 <= 160 * 1024, "linear_f32_smallm: K=%ld too large", (long)K);
  for (int64_t m0 = 0; m0 < M; m0 += max_rows) {
    const int Mc = (int)((M - m0) < max_rows ? (M - m0) : max_rows);
    const size_t lds = (size_t)Mc * K * 4;
    const int cols_per_wave = 4;
    const unsigned grid = (unsigned)((N + 4 * cols_per_wave - 1) / (4 * cols_per_wave));
    auto kern = linear_f32_smallm_kernel<16>;
    // (function-local static: initialised once, thread-safe)
    static const bool attr_ok = !(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess);
    if (!attr_ok) {
        lcv_set_error("linear_f32_smallm: cannot raise dynamic LDS");
        return LCV_EDEVICE;
    }
    hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, s, a + m0 * K, (const bf16_t*)w, (const bf16_t*)bias,
                       out + m0 * N, Mc, N, (int)K, act_in, cols_per_wave);
    LCV_LAUNCH_CHECK("linear_f32_smallm");
  }
  return LCV_OK;
}

// ---------------------------------------------------------------------------
// LoRA down-projection: h[M,Rpad] = bf16( s * bf16( x[M,K] A[R,K]^T ) ), zero padded to Rpad columns.
// Also serves the backward g = s * dy B (x := dy, A := B^T).  HBM-bound on x (rank r flop/B):
// one wave owns 4 rows so every 16-byte piece of A (L1/L2-resident, <= 800 KB) is reused 4 times.
// ---------------------------------------------------------------------------
template <int RMAX>
__global__ __launch_bounds__(256) void lora_down_kernel(const bf16_t* __restrict__ x,
                                                        const bf16_t* __restrict__ A,
                                                        bf16_t* __restrict__ hout, int64_t M, int K, int R,
                                                        int Rpad, int64_t ldx, float s) {
  const int lane = threadIdx.x & 63;
  const int64_t row0 = ((int64_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * 4;
  if (row0 >= M) return;
  int64_t rows[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) rows[i] = (row0 + i < M) ? row0 + i : M - 1;
  float acc[4][RMAX];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < RMAX; ++j) acc[i][j] = 0.f;
  for (int k = lane * 8; k < K; k += 512) {
    float xf[4][8];
#pragma unroll
    for (int i = 0; i < 4; ++i) unpack8(*reinterpret_cast<const u16x8*>(x + rows[i] * ldx + k), xf[i]);
#pragma unroll
    for (int j = 0; j < RMAX; ++j) {
      if (j < R) {
        float af[8];
        unpack8(*reinterpret_cast<const u16x8*>(A + (int64_t)j * K + k), af);
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int e = 0; e < 8; ++e) acc[i][j] += xf[i][e] * af[e];
      }
    }
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    float mine = 0.f;  // lane j keeps column j
#pragma unroll
    for (int j = 0; j < RMAX; ++j) {
      if (j < R) {
        const float t = wave_sum(acc[i][j]);
        if (lane == j) mine = t;
      }
    }
    if (row0 + i < M && lane < Rpad) hout[(row0 + i) * Rpad + lane] = (lane < R) ? f2bf(s * bfround(mine)) : (bf16_t)0;
  }
}

extern "C" int lcv_lora_down(const void* x, const void* A, void* h, int64_t M, int64_t K, int64_t R,
                             int64_t Rpad, int64_t ldx, float s, void* stream) {
  LCV_CHECK_ARG(x && A && h, "lora_down: null pointer");
  LCV_CHECK_ARG(R >= 1 && R <= 32 && Rpad >= R && Rpad <= 64, "lora_down: rank %ld unsupported (1..32, Rpad <= 64)", (long)R);
  LCV_CHECK_ARG(K % 8 == 0 && ldx % 8 == 0, "lora_down: K and ldx must be multiples of 8");
  if (M == 0) return LCV_OK;
  const unsigned blocks = (unsigned)((M + 15) / 16);
  hipStream_t st = (hipStream_t)stream;
  if (R <= 8)
    hipLaunchKernelGGL(lora_down_kernel<8>, dim3(blocks), dim3(256), 0, st, (const bf16_t*)x, (const bf16_t*)A, (bf16_t*)h, M, (int)K, (int)R, (int)Rpad, ldx, s);
  else if (R <= 16)
    hipLaunchKernelGGL(lora_down_kernel<16>, dim3(blocks), dim3(256), 0, st, (const bf16_t*)x, (const bf16_t*)A, (bf16_t*)h, M, (int)K, (int)R, (int)Rpad, ldx, s);
  else
    hipLaunchKernelGGL(lora_down_kernel<32>, dim3(blocks), dim3(256), 0, st, (const bf16_t*)x, (const bf16_t*)A, (bf16_t*)h, M, (int)K, (int)R, (int)Rpad, ldx, s);
  LCV_LAUNCH_CHECK("lora_down");
  return LCV_OK;
}
