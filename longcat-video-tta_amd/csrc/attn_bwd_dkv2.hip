// Attention backward, pass A (dK, dV), second form - for the self-attention path where q is pre-scaled into log2 units
// (scale * log2(e) == 1).  attn_bwd_dkv_kernel keeps K and V fragments AND both accumulators in registers, which needs ~290
// registers: one wave per SIMD (nothing hides its LDS / barrier stalls) and the compiler shuttles the score tiles through
// AGPRs (96 v_accvgpr moves per tile).  Here:
//   * workgroup = 4 waves = 128 keys, and TWO workgroups share a CU (2 waves per SIMD from different workgroups: their
//     barriers are independent, so one wave's LDS / barrier waits sit under the other's MFMAs);
//   * the K rows of the block live in LDS (32 KiB, loaded once by LDS-DMA) and are read as B operands per tile; only the V rows
//     stay in registers - 256 registers suffice (amdgpu_waves_per_eu(2, 2));
//   * Q and dO tiles (32 query rows) arrive by LDS-DMA, double-buffered;
//   * row constants as the initial accumulator: S starts at -lse (log2 units) and dP at -delta, read from LDS straight into the
//     accumulators, so P = exp2(S) and dS' = P * dP are 2 vector instructions per score; the softmax scale is applied to dK once
//     in the epilogue.
#include "lcv_common.h"
#include <stdlib.h>

typedef __attribute__((address_space(3))) unsigned char lds_u8;
typedef __attribute__((address_space(1))) void gbl_void_k;
typedef __attribute__((address_space(3))) void lds_void_k;
#define AS3 __attribute__((address_space(3)))

struct AttnBwdDkv2Params {
  const bf16_t* q;
  const bf16_t* k;
  const bf16_t* v;
  const bf16_t* d_o;
  const float* lse;
  const float* consts;   // per (b, h): nlse2[Nqp] | ndelta[Nqp] (attn_bwd_delta_kernel), Nqp = roundup(Nq, 32)
  bf16_t* dk;
  bf16_t* dv;
  int64_t Nq, Nk;
  int H;
  int64_t q_sb, q_sn, q_sh, k_sb, k_sn, k_sh, v_sb, v_sn, v_sh, o_sb, o_sn, o_sh;
  int64_t dk_sb, dk_sn, dk_sh, dv_sb, dv_sn, dv_sh;
  float scale;
  int accumulate_kv;
  int gx, xcd_remap;   // blocks per (batch, head); head-per-XCD block order (speed only: see attn_fwd.hip)
  int stagger;         // s_sleep argument (x 64 cycles) for every second resident workgroup; 0 = off
};

__device__ __forceinline__ int swz_k2(int row) { return ((row & 3) << 2) | ((row >> 2) & 3); }

// PIPE: the fragment reads of k-step ks+1 (and of transposed-read step j+1) are issued BEFORE the MFMAs of step ks (j), pinned
// with scheduling fences.  hipcc's own order is {3 reads; wait; 2 MFMAs} x 8 - every MFMA pair behind a full LDS latency,
// which the partner wave only half hides (matrix pipe 0.47 busy, profiles/r02_attn_pmc_summary.md).
// NW: waves per workgroup = 32 * NW keys per block.  NW = 4: two workgroups per CU (above).  NW = 8 (round 3, LCV_ATTN_BWD_DKV_WAVES=8): ONE
// 8-wave workgroup per CU owns 256 keys, so a Q / dO tile staged into LDS feeds twice the MFMAs: 16 KiB of LDS-DMA fill per 256
// MFMAs instead of per 128 (at the MFMA rate that is 8 instead of 16 B/clk per CU against the ~20-25 B/clk the fill path delivers).
template <bool PIPE, int NW = 4>
__global__ __launch_bounds__(NW * 64) __attribute__((amdgpu_waves_per_eu(2, 2))) void attn_bwd_dkv2_kernel(const AttnBwdDkv2Params p) {
  constexpr int QT = 32;
  constexpr int NP = 8 / NW;                            // one-KiB Q (and dO) pieces per wave and tile
  constexpr int TILE_BYTES = QT * 256;                  // one [32][128] bf16 tile
  constexpr int STAGE = 2 * TILE_BYTES + 2 * QT * 4;    // Q | dO | -lse | -delta
  constexpr int K_BYTES = NW * 32 * 256;                // the block's K rows
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  lds_u8* lds_k = (lds_u8*)smem;
  lds_u8* lds = lds_k + K_BYTES;                        // [2] stages

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 31, h = lane >> 5;
  // Block order (speed only): with the remap each XCD walks the key blocks of ITS OWN (batch, head) pairs, so that head's
  // Q / dO rows (re-streamed by every key block) stay in one 4 MiB L2 instead of being fetched into all eight
  int kb, head;
  int64_t b;
  if (p.xcd_remap) {
    const int id = blockIdx.x;
    const int xcd = id & 7, j = id >> 3;
    const int pair = (j / p.gx) * 8 + xcd;
    kb = j - (j / p.gx) * p.gx;
    head = pair % p.H;
    b = pair / p.H;
  } else {
    kb = blockIdx.x; head = blockIdx.y; b = blockIdx.z;
  }
  const int64_t key0 = (int64_t)kb * (NW * 32);
  const bf16_t* kbase = p.k + b * p.k_sb + (int64_t)head * p.k_sh;

  // ---- V rows of this lane's key as B operands (registers); K rows of the whole block -> LDS by DMA ----
  bf16x8 vf[8];
  {
    int64_t krow = key0 + wave * 32 + r;
    if (krow > p.Nk - 1) krow = p.Nk - 1;
    const bf16_t* vp = p.v + b * p.v_sb + krow * p.v_sn + (int64_t)head * p.v_sh + 8 * h;
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) vf[ks] = *reinterpret_cast<const bf16x8*>(vp + 16 * ks);
  }
#pragma unroll
  for (int i = 0; i < 8; ++i) {  // 32 one-KiB pieces (4 rows each), 8 per wave
    const int row = 4 * (8 * wave + i) + (lane >> 4);
    int64_t g = key0 + row;
    if (g > p.Nk - 1) g = p.Nk - 1;
    const int col = 8 * ((lane & 15) ^ swz_k2(row));
    __builtin_amdgcn_global_load_lds((gbl_void_k*)(kbase + g * p.k_sn + col), (lds_void_k*)(lds_k + (8 * wave + i) * 1024), 16, 0, 0);
  }

  // ---- Q / dO tile staging by DMA (2 + 2 pieces per wave), -lse / -delta through two registers of the first 32 threads ----
  const bf16_t* qbase = p.q + b * p.q_sb + (int64_t)head * p.q_sh;
  const bf16_t* dobase = p.d_o + b * p.o_sb + (int64_t)head * p.o_sh;
  const int64_t Nqp = (p.Nq + 31) / 32 * 32;
  const char* cbase_u = lcv_uniform_ptr(p.consts + (b * p.H + head) * 2 * Nqp);
  const unsigned coff = (unsigned)((lane < 32 ? lane : Nqp + lane - 32) * 4);   // one dword piece: 32 x nlse2 | 32 x ndelta
  int dma_row[NP];
  unsigned qoff[NP], dooff[NP];
#pragma unroll
  for (int i = 0; i < NP; ++i) {
    dma_row[i] = 4 * (NP * wave + i) + (lane >> 4);
    const int col = 8 * ((lane & 15) ^ swz_k2(dma_row[i]));
    qoff[i] = (unsigned)((dma_row[i] * p.q_sn + col) * 2);
    dooff[i] = (unsigned)((dma_row[i] * p.o_sn + col) * 2);
  }
  const char* qbase_u = lcv_uniform_ptr(qbase);
  const char* dobase_u = lcv_uniform_ptr(dobase);
  const unsigned stage_addr0 = (unsigned)(uintptr_t)lds;
  auto load_tile = [&](int64_t q0, int buf) {
    // asm-issued (lcv_common.h: a builtin DMA would be waited for before the next fragment read); waited for before the barrier.
    // Source = scalar tile base + a per-lane 32-bit offset that never changes; the ragged last tile clamps rows per lane.
    const unsigned sb = stage_addr0 + (unsigned)(buf * STAGE) + (unsigned)wave * (unsigned)(NP * 1024);
    const char* qt = qbase_u + q0 * (2 * p.q_sn);
    const char* dt = dobase_u + q0 * (2 * p.o_sn);
    if (q0 + QT <= p.Nq) {
#pragma unroll
      for (int i = 0; i < NP; ++i) {
        lcv_lds_dma16_sv(qoff[i], qt, sb + 1024u * i);
        lcv_lds_dma16_sv(dooff[i], dt, sb + (unsigned)TILE_BYTES + 1024u * i);
      }
    } else {
#pragma unroll
      for (int i = 0; i < NP; ++i) {
        int64_t back = q0 + dma_row[i] - (p.Nq - 1);
        if (back < 0) back = 0;
        lcv_lds_dma16(qt + qoff[i] - back * p.q_sn * 2, sb + 1024u * i);
        lcv_lds_dma16(dt + dooff[i] - back * p.o_sn * 2, sb + (unsigned)TILE_BYTES + 1024u * i);
      }
    }
    // the tile's row constants, already in accumulator form, by one dword piece (no register round trip, no compiler-visible
    // load in the loop: its wait would be a vmcnt(0) that also drains the pieces above)
    if (wave == 0) lcv_lds_dma4_sv(coff, cbase_u + q0 * 4, stage_addr0 + (unsigned)(buf * STAGE + 2 * TILE_BYTES));
  };

  // ---- LDS read addresses ----
  const int rf = swz_k2(r);
  const int row_off = 256 * r;
  const int krow_off = 256 * (32 * wave + r);
  const int q4 = (lane >> 2) & 3, p4 = lane & 3, g1 = (lane >> 4) & 1;
  int t_base[2], t_low[2];
#pragma unroll
  for (int half = 0; half < 2; ++half) {
    t_base[half] = 256 * (4 * h + 8 * half + q4) + 8 * (p4 & 1);
    t_low[half] = (2 * g1 + (p4 >> 1)) ^ (h + 2 * half);
  }

  f32x16 dkacc[4], dvacc[4];
#pragma unroll
  for (int d = 0; d < 4; ++d)
#pragma unroll
    for (int e = 0; e < 16; ++e) { dkacc[d][e] = 0.f; dvacc[d][e] = 0.f; }

  const int nt = (int)((p.Nq + QT - 1) / QT);
  if (p.stagger) {   // A/B knob: the second workgroup of a CU (dispatch order: id + 256) starts half a tile late
    const unsigned lin = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);
    if ((lin >> 8) & 1)
      for (int i = 0; i < p.stagger; ++i) __builtin_amdgcn_s_sleep(1);
  }
  load_tile(0, 0);
  lcv_dma_wait_all();
  __syncthreads();  // (the LDS-DMA of the K rows and of tile 0 has landed)

  for (int t = 0; t < nt; ++t) {
    const int buf = t & 1;
    const bool has_next = (t + 1 < nt);
    if (has_next) load_tile((int64_t)(t + 1) * QT, buf ^ 1);  // buf ^ 1 was last read in iteration t-1
    const lds_u8* qb = lds + buf * STAGE;
    const lds_u8* db = qb + TILE_BYTES;
    const lds_u8* lb = qb + 2 * TILE_BYTES;

    // ---- S = Q K^T - lse, dP = dO V^T - delta : rows = query (registers), cols = key (lane); the row constants are the
    // initial accumulators (element e <-> query (e & 3) + 8 (e >> 2) + 4 h)
    f32x16 s, dp;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const f32x4 l4 = *reinterpret_cast<const AS3 f32x4*>(lb + (8 * g + 4 * h) * 4);
      const f32x4 d4 = *reinterpret_cast<const AS3 f32x4*>(lb + QT * 4 + (8 * g + 4 * h) * 4);
#pragma unroll
      for (int e = 0; e < 4; ++e) { s[4 * g + e] = l4[e]; dp[4 * g + e] = d4[e]; }
    }
    if constexpr (PIPE) {
      bf16x8 aq[2], ad[2], kf[2];
      auto ld1 = [&](int ks, int st) {
        const int co = 16 * ((2 * ks + h) ^ rf);
        aq[st] = *reinterpret_cast<const AS3 bf16x8*>(qb + row_off + co);
        kf[st] = *reinterpret_cast<const AS3 bf16x8*>(lds_k + krow_off + co);
        ad[st] = *reinterpret_cast<const AS3 bf16x8*>(db + row_off + co);
      };
      ld1(0, 0);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int ks = 0; ks < 8; ++ks) {
        if (ks < 7) ld1(ks + 1, (ks + 1) & 1);
        s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(aq[ks & 1], kf[ks & 1], s, 0, 0, 0);
        dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ad[ks & 1], vf[ks], dp, 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
    } else {
#pragma unroll
      for (int ks = 0; ks < 8; ++ks) {
        const int co = 16 * ((2 * ks + h) ^ rf);
        const bf16x8 aq = *reinterpret_cast<const AS3 bf16x8*>(qb + row_off + co);
        const bf16x8 ad = *reinterpret_cast<const AS3 bf16x8*>(db + row_off + co);
        const bf16x8 kfr = *reinterpret_cast<const AS3 bf16x8*>(lds_k + krow_off + co);   // same swizzle term: (32 w + r) & 15 == r & 15
        s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(aq, kfr, s, 0, 0, 0);
        dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ad, vf[ks], dp, 0, 0, 0);
      }
    }
    // ---- P = exp2(S), dS' = P * dP (the scale goes to dK in the epilogue) ----
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      s[e] = __builtin_amdgcn_exp2f(s[e]);
      dp[e] = s[e] * dp[e];
    }
    bf16x8 pb[2], dsb[2];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      pb[0][j] = (__bf16)s[j];
      pb[1][j] = (__bf16)s[8 + j];
      dsb[0][j] = (__bf16)dp[j];
      dsb[1][j] = (__bf16)dp[8 + j];
    }
    // ---- dV^T += dO^T P ; dK^T += Q^T dS'  (A operands by transposed reads of the dO / Q tiles) ----
    if constexpr (PIPE) {
      s16x4 tlo[2], thi[2], tlo2[2], thi2[2];
      auto ld2 = [&](int j, int st) {   // step j = 4 ss + d
        const int ss = j >> 2, d = j & 3;
        const int dx = 64 * (d ^ q4);
        const int a0 = t_base[0] + 4096 * ss + dx + 16 * t_low[0];
        const int a1 = t_base[1] + 4096 * ss + dx + 16 * t_low[1];
        tlo[st] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((AS3 s16x4*)(db + a0));
        thi[st] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((AS3 s16x4*)(db + a1));
        tlo2[st] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((AS3 s16x4*)(qb + a0));
        thi2[st] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((AS3 s16x4*)(qb + a1));
      };
      ld2(0, 0);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        if (j < 7) ld2(j + 1, (j + 1) & 1);
        const int ss = j >> 2, d = j & 3, st = j & 1;
        const bf16x8 dof = __builtin_bit_cast(bf16x8, __builtin_shufflevector(tlo[st], thi[st], 0, 1, 2, 3, 4, 5, 6, 7));
        dvacc[d] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(dof, pb[ss], dvacc[d], 0, 0, 0);
        const bf16x8 qtf = __builtin_bit_cast(bf16x8, __builtin_shufflevector(tlo2[st], thi2[st], 0, 1, 2, 3, 4, 5, 6, 7));
        dkacc[d] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(qtf, dsb[ss], dkacc[d], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
    } else {
#pragma unroll
      for (int ss = 0; ss < 2; ++ss) {
#pragma unroll
        for (int d = 0; d < 4; ++d) {
          const int dx = 64 * (d ^ q4);
          const int a0 = t_base[0] + 4096 * ss + dx + 16 * t_low[0];
          const int a1 = t_base[1] + 4096 * ss + dx + 16 * t_low[1];
          const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((AS3 s16x4*)(db + a0));
          const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((AS3 s16x4*)(db + a1));
          const bf16x8 dof = __builtin_bit_cast(bf16x8, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
          dvacc[d] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(dof, pb[ss], dvacc[d], 0, 0, 0);
          const s16x4 lo2 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((AS3 s16x4*)(qb + a0));
          const s16x4 hi2 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((AS3 s16x4*)(qb + a1));
          const bf16x8 qtf = __builtin_bit_cast(bf16x8, __builtin_shufflevector(lo2, hi2, 0, 1, 2, 3, 4, 5, 6, 7));
          dkacc[d] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(qtf, dsb[ss], dkacc[d], 0, 0, 0);
        }
      }
    }
    if (has_next) lcv_dma_wait_all();   // the next tile, requested at the top of this one, has had the whole tile to land
    __syncthreads();
  }

  // ---- epilogue: acc[d][e] = dX^T[dim = 32 d + (e & 3) + 8 (e >> 2) + 4 h][key = lane & 31] ----
  const int64_t krow = key0 + wave * 32 + r;
  if (krow < p.Nk) {
    bf16_t* dkp = p.dk + b * p.dk_sb + krow * p.dk_sn + (int64_t)head * p.dk_sh;
    bf16_t* dvp = p.dv + b * p.dv_sb + krow * p.dv_sn + (int64_t)head * p.dv_sh;
#pragma unroll
    for (int d = 0; d < 4; ++d)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int col = 32 * d + 8 * i + 4 * h;
        float kv4[4], vv4[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) { kv4[e] = dkacc[d][4 * i + e] * p.scale; vv4[e] = dvacc[d][4 * i + e]; }
        if (p.accumulate_kv) {
          const u16x4 ok = *reinterpret_cast<const u16x4*>(dkp + col);
          const u16x4 ov = *reinterpret_cast<const u16x4*>(dvp + col);
#pragma unroll
          for (int e = 0; e < 4; ++e) { kv4[e] += bf2f(ok[e]); vv4[e] += bf2f(ov[e]); }
        }
        u16x4 pk, pv;
#pragma unroll
        for (int e = 0; e < 4; ++e) { pk[e] = f2bf(kv4[e]); pv[e] = f2bf(vv4[e]); }
        *reinterpret_cast<u16x4*>(dkp + col) = pk;
        *reinterpret_cast<u16x4*>(dvp + col) = pv;
      }
  }
}

// called by lcv_attn_bwd (attn_bwd.hip) when scale * log2(e) == 1 and LCV_ATTN_BWD_VAR allows it
int attn_bwd_dkv2_launch(const void* q, const void* k, const void* v, const void* d_o, const float* lse, const float* delta,
                         void* dk, void* dv, int accumulate_kv, int64_t B, int64_t H, int64_t Nq, int64_t Nk, int64_t q_sb,
                         int64_t q_sn, int64_t q_sh, int64_t k_sb, int64_t k_sn, int64_t k_sh, int64_t v_sb, int64_t v_sn,
                         int64_t v_sh, int64_t o_sb, int64_t o_sn, int64_t o_sh, int64_t dk_sb, int64_t dk_sn, int64_t dk_sh,
                         int64_t dv_sb, int64_t dv_sn, int64_t dv_sh, float scale, hipStream_t s) {
  AttnBwdDkv2Params p;
  p.q = (const bf16_t*)q; p.k = (const bf16_t*)k; p.v = (const bf16_t*)v; p.d_o = (const bf16_t*)d_o;
  p.lse = lse; p.consts = delta; p.dk = (bf16_t*)dk; p.dv = (bf16_t*)dv; p.Nq = Nq; p.Nk = Nk; p.H = (int)H;
  p.q_sb = q_sb; p.q_sn = q_sn; p.q_sh = q_sh; p.k_sb = k_sb; p.k_sn = k_sn; p.k_sh = k_sh;
  p.v_sb = v_sb; p.v_sn = v_sn; p.v_sh = v_sh; p.o_sb = o_sb; p.o_sn = o_sn; p.o_sh = o_sh;
  p.dk_sb = dk_sb; p.dk_sn = dk_sn; p.dk_sh = dk_sh; p.dv_sb = dv_sb; p.dv_sn = dv_sn; p.dv_sh = dv_sh;
  p.scale = scale; p.accumulate_kv = accumulate_kv;
  const char* we = lcv_knob("LCV_ATTN_BWD_DKV_WAVES");   // A/B knob: 4 = two 4-wave workgroups per CU (128 keys each), 8 = one 8-wave (256 keys)
  const int nw = (we && we[0] == '8') ? 8 : 4;
  const size_t lds = (size_t)nw * 32 * 256 + 2 * (2 * 32 * 256 + 2 * 32 * 4);
    const int l4 = 4 * 32 * 256 + 2 * (2 * 32 * 256 + 2 * 32 * 4), l8 = 8 * 32 * 256 + 2 * (2 * 32 * 256 + 2 * 32 * 4);
  // (function-local static: initialised once, thread-safe)
  static const bool attr_ok = !(hipFuncSetAttribute((const void*)attn_bwd_dkv2_kernel<true, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, l4) != hipSuccess ||
        hipFuncSetAttribute((const void*)attn_bwd_dkv2_kernel<false, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, l4) != hipSuccess ||
        hipFuncSetAttribute((const void*)attn_bwd_dkv2_kernel<true, 8>, hipFuncAttributeMaxDynamicSharedMemorySize, l8) != hipSuccess);
  if (!attr_ok) {
      lcv_set_error("attn_bwd: cannot raise dynamic LDS");
      return LCV_EDEVICE;
  }
  { const char* se = lcv_knob("LCV_ATTN_BWD_STAGGER"); p.stagger = se ? atoi(se) : 0; if (p.stagger < 0 || p.stagger > 127) p.stagger = 0; }
  const char* pe = lcv_knob("LCV_ATTN_BWD_PIPE");   // A/B knob: 0 = hipcc's own read / MFMA order
  const bool pipe = !(pe && pe[0] == '0');
  const unsigned gx = (unsigned)((Nk + nw * 32 - 1) / (nw * 32));
  // A/B knob LCV_ATTN_BWD_XCD=1 enables the head-per-XCD block order.  OFF by default: at the K3-TTA shapes (25 200 keys x 32
  // heads) it measured 27.06 vs 26.51 ms per layer in one process - unlike the forward, these passes are not helped by it
  const char* xe = lcv_knob("LCV_ATTN_BWD_XCD");
  p.gx = (int)gx;
  p.xcd_remap = ((B * H) % 8 == 0 && gx >= 8 && xe && xe[0] == '1') ? 1 : 0;
  const dim3 grid = p.xcd_remap ? dim3(gx * (unsigned)(H * B)) : dim3(gx, (unsigned)H, (unsigned)B);
  if (nw == 8) hipLaunchKernelGGL((attn_bwd_dkv2_kernel<true, 8>), grid, dim3(512), lds, s, p);
  else if (pipe) hipLaunchKernelGGL((attn_bwd_dkv2_kernel<true, 4>), grid, dim3(256), lds, s, p);
  else hipLaunchKernelGGL((attn_bwd_dkv2_kernel<false, 4>), grid, dim3(256), lds, s, p);
  LCV_LAUNCH_CHECK("attn_bwd_dkv2");
  return LCV_OK;
}
