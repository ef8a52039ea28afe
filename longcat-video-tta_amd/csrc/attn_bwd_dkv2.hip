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
  const float* delta;
  bf16_t* dk;
  bf16_t* dv;
  int64_t Nq, Nk;
  int H;
  int64_t q_sb, q_sn, q_sh, k_sb, k_sn, k_sh, v_sb, v_sn, v_sh, o_sb, o_sn, o_sh;
  int64_t dk_sb, dk_sn, dk_sh, dv_sb, dv_sn, dv_sh;
  float scale;
  int accumulate_kv;
  int gx, xcd_remap;   // blocks per (batch, head); head-per-XCD block order (speed only: see attn_fwd.hip)
};

__device__ __forceinline__ int swz_k2(int row) { return ((row & 3) << 2) | ((row >> 2) & 3); }

__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void attn_bwd_dkv2_kernel(const AttnBwdDkv2Params p) {
  constexpr int QT = 32;
  constexpr int TILE_BYTES = QT * 256;                  // one [32][128] bf16 tile
  constexpr int STAGE = 2 * TILE_BYTES + 2 * QT * 4;    // Q | dO | -lse | -delta
  constexpr int K_BYTES = 128 * 256;                    // the block's K rows
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
  const int64_t key0 = (int64_t)kb * 128;
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
  const float* lsebase = p.lse + (b * p.H + head) * p.Nq;
  const float* delbase = p.delta + (b * p.H + head) * p.Nq;
  int dma_row[2], dma_col[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    dma_row[i] = 8 * wave + 4 * i + (lane >> 4);
    dma_col[i] = 8 * ((lane & 15) ^ swz_k2(dma_row[i]));
  }
  float lreg = 0.f, dlreg = 0.f;
  auto load_tile = [&](int64_t q0, int buf) {
    lds_u8* sb = lds + buf * STAGE + wave * 2048;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      int64_t row = q0 + dma_row[i];
      if (row > p.Nq - 1) row = p.Nq - 1;
      __builtin_amdgcn_global_load_lds((gbl_void_k*)(qbase + row * p.q_sn + dma_col[i]), (lds_void_k*)(sb + 1024 * i), 16, 0, 0);
      __builtin_amdgcn_global_load_lds((gbl_void_k*)(dobase + row * p.o_sn + dma_col[i]), (lds_void_k*)(sb + TILE_BYTES + 1024 * i),
                                       16, 0, 0);
    }
    if (tid < QT) {
      const int64_t row = q0 + tid;
      // rows past Nq: -lse = -inf makes their P (and dS) exactly zero
      lreg = (row < p.Nq) ? -lsebase[row] * 1.4426950408889634f : -INFINITY;
      dlreg = (row < p.Nq) ? -delbase[row] : 0.f;
    }
  };
  auto store_consts = [&](int buf) {
    if (tid < QT) {
      lds_u8* sb = lds + buf * STAGE + 2 * TILE_BYTES;
      *reinterpret_cast<AS3 float*>(sb + tid * 4) = lreg;
      *reinterpret_cast<AS3 float*>(sb + QT * 4 + tid * 4) = dlreg;
    }
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
  load_tile(0, 0);
  store_consts(0);
  __syncthreads();  // (drains the LDS-DMA of the K rows and of tile 0: vmcnt 0)

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
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) {
      const int co = 16 * ((2 * ks + h) ^ rf);
      const bf16x8 aq = *reinterpret_cast<const AS3 bf16x8*>(qb + row_off + co);
      const bf16x8 ad = *reinterpret_cast<const AS3 bf16x8*>(db + row_off + co);
      const bf16x8 kfr = *reinterpret_cast<const AS3 bf16x8*>(lds_k + krow_off + co);   // same swizzle term: (32 w + r) & 15 == r & 15
      s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(aq, kfr, s, 0, 0, 0);
      dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ad, vf[ks], dp, 0, 0, 0);
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
    if (has_next) store_consts(buf ^ 1);
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
  p.lse = lse; p.delta = delta; p.dk = (bf16_t*)dk; p.dv = (bf16_t*)dv; p.Nq = Nq; p.Nk = Nk; p.H = (int)H;
  p.q_sb = q_sb; p.q_sn = q_sn; p.q_sh = q_sh; p.k_sb = k_sb; p.k_sn = k_sn; p.k_sh = k_sh;
  p.v_sb = v_sb; p.v_sn = v_sn; p.v_sh = v_sh; p.o_sb = o_sb; p.o_sn = o_sn; p.o_sh = o_sh;
  p.dk_sb = dk_sb; p.dk_sn = dk_sn; p.dk_sh = dk_sh; p.dv_sb = dv_sb; p.dv_sn = dv_sn; p.dv_sh = dv_sh;
  p.scale = scale; p.accumulate_kv = accumulate_kv;
  const size_t lds = 128 * 256 + 2 * (2 * 32 * 256 + 2 * 32 * 4);
  static bool attr_set = false;
  if (!attr_set) {
    if (hipFuncSetAttribute((const void*)attn_bwd_dkv2_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) {
      lcv_set_error("attn_bwd: cannot raise dynamic LDS");
      return LCV_EDEVICE;
    }
    attr_set = true;
  }
  const unsigned gx = (unsigned)((Nk + 127) / 128);
  // A/B knob LCV_ATTN_BWD_XCD=1 enables the head-per-XCD block order.  OFF by default: at the K3-TTA shapes (25 200 keys x 32
  // heads) it measured 27.06 vs 26.51 ms per layer in one process - unlike the forward, these passes are not helped by it
  const char* xe = getenv("LCV_ATTN_BWD_XCD");
  p.gx = (int)gx;
  p.xcd_remap = ((B * H) % 8 == 0 && gx >= 8 && xe && xe[0] == '1') ? 1 : 0;
  const dim3 grid = p.xcd_remap ? dim3(gx * (unsigned)(H * B)) : dim3(gx, (unsigned)H, (unsigned)B);
  hipLaunchKernelGGL(attn_bwd_dkv2_kernel, grid, dim3(256), lds, s, p);
  LCV_LAUNCH_CHECK("attn_bwd_dkv2");
  return LCV_OK;
}
