// Flash attention forward for the LongCat-Video DiT (dense, non-causal, D = 128, bf16 in,
// fp32 accumulate).  One kernel serves 3-D self-attention (Nq = Nk = T*h*w), the
// conditioning split (cond-q x cond-kv, noise-q x all-kv), the KV-cached denoise step and
// the text cross-attention (Nk <= 512) through strides and pointer offsets.
//
// gfx950 structure
//   * workgroup = 8 waves = 256 query rows of one (batch, head); each wave owns 32 rows.
//     KV tile = 64 keys, K and V double-buffered in LDS (2 x 32 KiB), register-staged
//     prefetch of tile t+1 issued before the MFMA work of tile t, one barrier per tile.
//   * scores are computed TRANSPOSED, S^T = K Q^T (v_mfma_f32_32x32x16_bf16, A = K rows from
//     LDS via ds_read_b128, B = Q held in registers).  The accumulator then has the query
//     on the lane and the keys in the registers, so the softmax row max / row sum / rescale
//     are lane-local (one cross-half exchange each), and the exponentiated tile is already
//     the B operand of O^T += V^T P^T — P never touches LDS.
//   * V^T fragments come from the row-major V tile with ds_read_b64_tr_b16; K and V share
//     one XOR-swizzled 256-byte-row image that is conflict-free for both read kinds.
//   * algorithmic work per launch: 4*Nq*Nk*128 flop and (Nq*2 + Nk*2)*128*2 bytes per (b, h).
#include "lcv_common.h"
#include <type_traits>

typedef __attribute__((address_space(3))) unsigned char lds_u8;

struct AttnFwdParams {
  const bf16_t* q;
  const bf16_t* k;
  const bf16_t* v;
  bf16_t* o;
  float* lse;
  int64_t Nq, Nk;
  int H;
  int64_t q_sb, q_sn, q_sh, k_sb, k_sn, k_sh, v_sb, v_sn, v_sh, o_sb, o_sn, o_sh;
  float scale, scale_log2e;
  int gx, xcd_remap;  // q-blocks per (b,h); head-per-XCD block order when (B*H) % 8 == 0
};

#define RESCALE_THR 6.0f  // log2 units: the running max may lag by up to 2^6 before O and l are rescaled

// exchange with the partner lane (l ^ 32) by ONE v_permlane32_swap: r[0] = low-half values, r[1] = high-half values in
// every lane.  The two operands must be distinct registers (the instruction swaps halves BETWEEN them; the compiler
// folds identical operands into one register and the swap degenerates), hence the opaque copy.
__device__ __forceinline__ void half_pair(float v, float& lo, float& hi) {
  // inline asm on purpose: hipcc 7.2 folds the two results of __builtin_amdgcn_permlane32_swap into one value when
  // both operands derive from the same variable (observed: `lo + hi` became `lo + lo`).  The leading s_nop 1 covers
  // the "VALU write -> v_permlane read" hazard (2 wait states) that the compiler does not pad inside an asm string.
  float a = v, b = v;
  asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 0" : "+v"(a), "+v"(b));
  lo = a;  // [low-half value | low-half value]
  hi = b;  // [high-half value | high-half value]
}
__device__ __forceinline__ float half_max(float v) {
  float lo, hi;
  half_pair(v, lo, hi);
  return fmaxf(lo, hi);
}
__device__ __forceinline__ float half_sum(float v) {
  float lo, hi;
  half_pair(v, lo, hi);
  return lo + hi;
}

// byte offset of 16-byte chunk `ch` (0..15) of row `row` in a [rows][128] bf16 tile
__device__ __forceinline__ int tile_off(int row, int ch) {
  return 256 * row + 16 * (ch ^ (((row & 3) << 2) | ((row >> 2) & 3)));
}

// XATTN only names the instantiation used for the short-KV text cross-attention (Nk <= 512) so that profiles list it
// separately from the self-attention launches (the dominant kernel); the code path is the same.
//
// VAR bit 0: K/V tiles go global -> LDS by LDS-DMA (global_load_lds_dwordx4; the swizzle is applied on the SOURCE column,
//            the destination is lane-linear) from per-lane pointers that advance by one tile per iteration: no staging
//            registers, no ds_write, no per-tile 64-bit address arithmetic.
//            (+3.6 % at N = 46 800.  Tried and dropped: row sums on the matrix core with an all-ones A operand, -3.6 %;
//            a three-stage {QK^T | softmax | PV} ping-pong with waves 4-7 one stage behind, raw barriers and counted
//            vmcnt: -1 %; the 16x16x32 MFMA shape: -10 % (this loop is vector-issue bound and that shape doubles the
//            MFMA issue slots).  Sources of the dropped variants: scratch/tried/.)
typedef __attribute__((address_space(1))) void gbl_void_t;
typedef __attribute__((address_space(3))) void lds_void_t;

template <int NWAVES, int PRIO, bool XATTN, int VAR>
__global__ __launch_bounds__(NWAVES * 64) void attn_fwd_kernel(const AttnFwdParams p) {
  constexpr bool DMA = (VAR & 1) != 0;
  // VAR bit 1: the caller pre-scaled Q so that Q.K is already the exponent in log2 units (scale * log2(e) == 1: the
  // self-attention path folds head_dim^-0.5 * log2(e) into the q RMSNorm/RoPE kernel before ITS bf16 rounding).  The score
  // accumulators then START at -running_max (a 16-register vector that changes only on a rescale) instead of 0, and
  // P = exp2(accumulator) needs no per-element multiply-subtract: 31 fewer vector instructions per tile in a loop whose
  // vector instructions do not hide behind its MFMAs.
  constexpr bool UNIT = (VAR & 2) != 0;
  static_assert(!DMA || NWAVES == 8, "LDS-DMA staging is laid out for 8 waves (2 K + 2 V instructions per wave)");
  constexpr int NT = NWAVES * 64;
  constexpr int QROWS = NWAVES * 32;
  constexpr int NCH = 1024 / NT;  // 16-byte chunks per thread per 64x128 tile
  constexpr int TILE_BYTES = 64 * 256;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  lds_u8* lds = (lds_u8*)smem;  // [2][K tile | V tile]

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 31, h = lane >> 5;
  // Block order (speed only, never correctness): workgroup ids are dealt round-robin over the 8 XCDs, so with the
  // remap every XCD walks the q-blocks of ITS OWN (batch, head) pairs and that head's K/V (24 MB at K3) streams
  // through one 4 MiB L2 instead of eight.
  int qb, head;
  int64_t b;
  if (p.xcd_remap) {
    const int id = blockIdx.x;
    const int xcd = id & 7, j = id >> 3;
    const int pair = (j / p.gx) * 8 + xcd;
    qb = j - (j / p.gx) * p.gx;
    head = pair % p.H;
    b = pair / p.H;
  } else {
    qb = blockIdx.x; head = blockIdx.y; b = blockIdx.z;
  }
  const int64_t q0 = (int64_t)qb * QROWS + wave * 32;

  const bf16_t* kbase = p.k + b * p.k_sb + (int64_t)head * p.k_sh;
  const bf16_t* vbase = p.v + b * p.v_sb + (int64_t)head * p.v_sh;

  // ---- Q fragments (B operand): lane holds Q[q0 + r][16*ks + 8*h .. +8] ----
  bf16x8 qf[8];
  {
    int64_t qrow = q0 + r;
    if (qrow > p.Nq - 1) qrow = p.Nq - 1;
    const bf16_t* qp = p.q + b * p.q_sb + qrow * p.q_sn + (int64_t)head * p.q_sh + 8 * h;
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) qf[ks] = *reinterpret_cast<const bf16x8*>(qp + 16 * ks);
  }

  // ---- staging roles: thread copies chunks c = tid + i*NT (row = c>>4, ch = c&15) ----
  int st_off[NCH];
  int st_row[NCH];
#pragma unroll
  for (int i = 0; i < NCH; ++i) {
    const int c = tid + i * NT;
    st_row[i] = c >> 4;
    st_off[i] = tile_off(c >> 4, c & 15);
  }
  const int st_col = (tid & 15) * 8;
  u32x4 kreg[NCH], vreg[NCH];
  // LDS-DMA roles: wave w fills rows 8 w .. 8 w + 7 of both tiles with 2 + 2 one-KiB instructions; lane l of instruction i
  // lands at row 8 w + 4 i + (l >> 4), physical chunk l & 15, which holds logical chunk (l & 15) ^ swizzle(row)
  const bf16_t* kdma[2];
  const bf16_t* vdma[2];
  int dma_row[2], dma_col[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    dma_row[i] = 8 * wave + 4 * i + (lane >> 4);
    dma_col[i] = 8 * ((lane & 15) ^ (((dma_row[i] & 3) << 2) | ((dma_row[i] >> 2) & 3)));
    kdma[i] = kbase + dma_row[i] * p.k_sn + dma_col[i];
    vdma[i] = vbase + dma_row[i] * p.v_sn + dma_col[i];
  }
  // issue the K (which = 0) or V (which = 1) rows of tile t into buffer buf; kdma / vdma point at this lane's rows of tile t
  // FULL: the caller knows tile t is a full 64-key tile (every tile but the last) - no ragged-row branch in its stream
  auto dma_half = [&](auto which_c, int t, int buf, auto full_c) {
    constexpr int which = decltype(which_c)::value;
    constexpr bool FULL = decltype(full_c)::value;
    lds_u8* dst = lds + buf * 2 * TILE_BYTES + which * TILE_BYTES + wave * 2048;
    const bf16_t** src = which ? vdma : kdma;
    const int64_t sn = which ? p.v_sn : p.k_sn;
    if (FULL || (int64_t)t * 64 + 64 <= p.Nk) {
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        __builtin_amdgcn_global_load_lds((gbl_void_t*)src[i], (lds_void_t*)(dst + 1024 * i), 16, 0, 0);
        src[i] += 64 * sn;
      }
    } else {  // ragged last tile: rows past Nk re-read the last key (masked below)
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        int64_t back = (int64_t)t * 64 + dma_row[i] - (p.Nk - 1);
        if (back < 0) back = 0;
        __builtin_amdgcn_global_load_lds((gbl_void_t*)(src[i] - back * sn), (lds_void_t*)(dst + 1024 * i), 16, 0, 0);
      }
    }
  };
  using K_ = std::integral_constant<int, 0>;
  using V_ = std::integral_constant<int, 1>;
  auto dma_tile = [&](int t, int buf, auto full_c) {
    dma_half(K_{}, t, buf, full_c);
    dma_half(V_{}, t, buf, full_c);
  };
  auto load_tile = [&](int64_t kv0) {
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
      int64_t row = kv0 + st_row[i];
      if (row > p.Nk - 1) row = p.Nk - 1;  // tail keys re-read the last row; they are masked below
      kreg[i] = *reinterpret_cast<const u32x4*>(kbase + row * p.k_sn + st_col);
      vreg[i] = *reinterpret_cast<const u32x4*>(vbase + row * p.v_sn + st_col);
    }
  };
  auto store_tile = [&](int buf) {
    lds_u8* kb = lds + buf * 2 * TILE_BYTES;
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
      *reinterpret_cast<__attribute__((address_space(3))) u32x4*>(kb + st_off[i]) = kreg[i];
      *reinterpret_cast<__attribute__((address_space(3))) u32x4*>(kb + TILE_BYTES + st_off[i]) = vreg[i];
    }
  };

  // ---- per-lane LDS read addresses ----
  const int kf = ((r & 3) << 2) | ((r >> 2) & 3);  // swizzle term of rows r and 32 + r
  const int k_row_off = 256 * r;
  // transposed V reads: lane = 16*g + 4*q4 + p4 ; supplies row q4, columns 4*p4..4*p4+3 of its block
  const int q4 = (lane >> 2) & 3, p4 = lane & 3, g1 = (lane >> 4) & 1;
  int v_base[2], v_low[2];
#pragma unroll
  for (int half = 0; half < 2; ++half) {
    v_base[half] = 256 * (4 * h + 8 * half + q4) + 8 * (p4 & 1);
    v_low[half] = (2 * g1 + (p4 >> 1)) ^ (h + 2 * half);
  }

  f32x16 oacc[4];
#pragma unroll
  for (int d = 0; d < 4; ++d)
#pragma unroll
    for (int e = 0; e < 16; ++e) oacc[d][e] = 0.f;
  float m_run = UNIT ? 0.f : -INFINITY;  // running max of the raw (unscaled) scores of this lane's query
  f32x16 minit;                           // UNIT: -m_run in every element = the C operand of each tile's first score MFMA
#pragma unroll
  for (int e = 0; e < 16; ++e) minit[e] = 0.f;
  float l_run = 0.f;        // this lane's partial row sum (its 32 of every 64 keys)

  if (PRIO == 1 && wave >= NWAVES / 2) __builtin_amdgcn_s_setprio(1);  // static priority for the younger half
  const int nt = (int)((p.Nk + 63) / 64);
  {
    if (DMA) {
      dma_tile(0, 0, std::false_type{});
    } else {
      load_tile(0);
      store_tile(0);
    }
    __syncthreads();  // (drains this wave's LDS-DMA: vmcnt 0)
  }

  // LAST = the final (possibly ragged) tile, peeled so that the steady-state body carries neither the mask code nor its
  // register copies, and `has_next` is a compile-time constant in both
  // NEXT_FULL: tile t+1 is known not to be the last one (no ragged-row code in the steady-state bodies)
  auto tile_body = [&](const int t, auto buf_c, auto last_c, auto next_full_c) {
    constexpr int buf = decltype(buf_c)::value;  // compile-time buffer: every LDS address is a fixed register + immediate
    constexpr bool has_next = !decltype(last_c)::value;
    if (has_next) {
      if (DMA) dma_tile(t + 1, buf ^ 1, next_full_c);  // buf ^ 1 was last read in iteration t-1; every wave has passed its barrier
      else load_tile((int64_t)(t + 1) * 64);
    }

    const lds_u8* kb = lds + buf * 2 * TILE_BYTES;
    const lds_u8* vb = kb + TILE_BYTES;

    // ---- S^T = K Q^T : two 32-key blocks ----
    f32x16 s0, s1;
#pragma unroll
    for (int e = 0; e < 16; ++e) { s0[e] = UNIT ? minit[e] : 0.f; s1[e] = UNIT ? minit[e] : 0.f; }
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) {
      const int co = 16 * ((2 * ks + h) ^ kf);
      const bf16x8 a0 = *reinterpret_cast<const __attribute__((address_space(3))) bf16x8*>(kb + k_row_off + co);
      const bf16x8 a1 = *reinterpret_cast<const __attribute__((address_space(3))) bf16x8*>(kb + 32 * 256 + k_row_off + co);
      s0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, qf[ks], s0, 0, 0, 0);
      s1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, qf[ks], s1, 0, 0, 0);
    }

    // ---- mask keys past Nk (last tile only; wave-uniform branch) ----
    if (!has_next && (p.Nk & 63)) {
      const int valid = (int)(p.Nk - (int64_t)t * 64);
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int key = (e & 3) + 8 * (e >> 2) + 4 * h;
        if (key >= valid) s0[e] = -INFINITY;
        if (key + 32 >= valid) s1[e] = -INFINITY;
      }
    }

    // ---- online softmax, lane-local except one cross-half exchange (v_permlane32_swap, no LDS round trip) ----
    // two independent max chains; built with -fno-honor-nans -mno-amdgpu-ieee (see lcv_hip/build.py) so fmaxf lowers to
    // bare v_max3_f32 without a canonicalising v_max per MFMA output (scores are never NaN: inputs are finite)
    float mxa = s0[0], mxb = s1[0];
#pragma unroll
    for (int e = 1; e < 16; ++e) {
      mxa = fmaxf(mxa, s0[e]);
      mxb = fmaxf(mxb, s1[e]);
    }
    float mx = fmaxf(mxa, mxb);
    mx = half_max(mx);
    // defer-max: rescale O / l only when some query's running max grows by more than 2^RESCALE_THR; otherwise keep the
    // old max (P <= 2^RESCALE_THR, harmless in fp32 sums and bf16 P).  The decision is wave-uniform and taken before
    // this tile's P exists and after the previous tile's PV finished, so nothing is ever scaled twice or not at all.
    float psum = 0.f;
    if constexpr (UNIT) {
      // the scores are already relative to the running max: mx > 0 is growth.  Tile 0 always takes the branch and sets
      // the first real maximum (m_run starts at 0 with O = l = 0, so its alpha multiplies zeros).
      if (__builtin_amdgcn_ballot_w64(mx > RESCALE_THR) != 0ull || t == 0) {
        const float d = (t == 0) ? mx : fmaxf(mx, 0.f);
        const float alpha = __builtin_amdgcn_exp2f(-d);
        m_run += d;
        l_run *= alpha;
#pragma unroll
        for (int dd = 0; dd < 4; ++dd)
#pragma unroll
          for (int e = 0; e < 16; ++e) oacc[dd][e] *= alpha;
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          s0[e] -= d;
          s1[e] -= d;
          minit[e] = -m_run;
        }
      }
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        s0[e] = __builtin_amdgcn_exp2f(s0[e]);
        s1[e] = __builtin_amdgcn_exp2f(s1[e]);
        psum += s0[e] + s1[e];
      }
    } else {
      if (__builtin_amdgcn_ballot_w64((mx - m_run) * p.scale_log2e > RESCALE_THR) != 0ull) {
        const float m_new = fmaxf(m_run, mx);
        const float alpha = __builtin_amdgcn_exp2f((m_run - m_new) * p.scale_log2e);  // first tile: exp2(-inf) = 0
        m_run = m_new;
        l_run *= alpha;
#pragma unroll
        for (int d = 0; d < 4; ++d)
#pragma unroll
          for (int e = 0; e < 16; ++e) oacc[d][e] *= alpha;
      }
      const float mc = m_run * p.scale_log2e;
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        s0[e] = __builtin_amdgcn_exp2f(s0[e] * p.scale_log2e - mc);
        s1[e] = __builtin_amdgcn_exp2f(s1[e] * p.scale_log2e - mc);
        psum += s0[e] + s1[e];
      }
    }
    l_run += psum;

    // ---- P^T as B operand: k-step kk covers keys 16*kk .. 16*kk+15 (permuted inside the step) ----
    bf16x8 pb[4];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      pb[0][j] = (__bf16)s0[j];
      pb[1][j] = (__bf16)s0[8 + j];
      pb[2][j] = (__bf16)s1[j];
      pb[3][j] = (__bf16)s1[8 + j];
    }

    // ---- O^T += V^T P^T ----
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
#pragma unroll
      for (int d = 0; d < 4; ++d) {
        const int dx = 64 * (d ^ q4);
        const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
            (__attribute__((address_space(3))) s16x4*)(vb + v_base[0] + 4096 * kk + dx + 16 * v_low[0]));
        const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
            (__attribute__((address_space(3))) s16x4*)(vb + v_base[1] + 4096 * kk + dx + 16 * v_low[1]));
        // join the two 64-bit results as whole registers (an element-wise short->bf16 rebuild is
        // miscompiled by hipcc 7.2 into a broadcast of element 0)
        const bf16x8 vf = __builtin_bit_cast(bf16x8, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
        oacc[d] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, pb[kk], oacc[d], 0, 0, 0);
      }
    }

    if (has_next && !DMA) store_tile(buf ^ 1);
    __syncthreads();
  };
  {
    using B0 = std::integral_constant<int, 0>;
    using B1 = std::integral_constant<int, 1>;
    using Y = std::true_type;
    using N = std::false_type;
    int t = 0;
    for (; t + 3 < nt; t += 2) {  // tiles t+1, t+2 <= nt-2: both full
      tile_body(t, B0{}, N{}, Y{});
      tile_body(t + 1, B1{}, N{}, Y{});
    }
    const int left = nt - t;  // 1, 2 or 3 tiles, starting on buffer 0
    if (left == 3) {
      tile_body(t, B0{}, N{}, Y{});
      tile_body(t + 1, B1{}, N{}, N{});
      tile_body(t + 2, B0{}, Y{}, N{});
    } else if (left == 2) {
      tile_body(t, B0{}, N{}, N{});
      tile_body(t + 1, B1{}, Y{}, N{});
    } else {
      tile_body(t, B0{}, Y{}, N{});
    }
  }

  // ---- epilogue ----
  const float l_tot = half_sum(l_run);
  const float inv = 1.0f / l_tot;
  const int64_t qrow = q0 + r;
  if (qrow < p.Nq) {
    bf16_t* op = p.o + b * p.o_sb + qrow * p.o_sn + (int64_t)head * p.o_sh;
#pragma unroll
    for (int d = 0; d < 4; ++d)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        u16x4 pk;
#pragma unroll
        for (int e = 0; e < 4; ++e) pk[e] = f2bf(oacc[d][4 * i + e] * inv);
        *reinterpret_cast<u16x4*>(op + 32 * d + 8 * i + 4 * h) = pk;
      }
    if (p.lse && h == 0) p.lse[(b * p.H + head) * p.Nq + qrow] = m_run * p.scale + __logf(l_tot);
  }
}

int attn_fwd_w64_launch(const void* q, const void* k, const void* v, void* o, float* lse, int64_t B, int64_t H, int64_t Nq,
                        int64_t Nk, int64_t q_sb, int64_t q_sn, int64_t q_sh, int64_t k_sb, int64_t k_sn, int64_t k_sh,
                        int64_t v_sb, int64_t v_sn, int64_t v_sh, int64_t o_sb, int64_t o_sn, int64_t o_sh, float scale,
                        int xcd_ok, hipStream_t s);
int attn_fwd_pipe_launch(const void* q, const void* k, const void* v, void* o, float* lse, int64_t B, int64_t H, int64_t Nq,
                         int64_t Nk, int64_t q_sb, int64_t q_sn, int64_t q_sh, int64_t k_sb, int64_t k_sn, int64_t k_sh,
                         int64_t v_sb, int64_t v_sn, int64_t v_sh, int64_t o_sb, int64_t o_sn, int64_t o_sh, float scale,
                         int xcd_ok, hipStream_t s);   // attn_fwd_pipe.hip

static thread_local const char* g_last_attn_kernel = "none";
extern "C" const char* lcv_attn_fwd_last_kernel(void) { return g_last_attn_kernel; }

extern "C" int lcv_attn_fwd(const void* q, const void* k, const void* v, void* o, float* lse, int64_t B,
                            int64_t H, int64_t Nq, int64_t Nk, int64_t q_sb, int64_t q_sn, int64_t q_sh,
                            int64_t k_sb, int64_t k_sn, int64_t k_sh, int64_t v_sb, int64_t v_sn,
                            int64_t v_sh, int64_t o_sb, int64_t o_sn, int64_t o_sh, float scale,
                            void* stream) {
  LCV_CHECK_ARG(q && k && v && o, "attn_fwd: null pointer");
  LCV_CHECK_ARG(B > 0 && H > 0 && H <= 65535 && B <= 65535, "attn_fwd: bad B/H");
  LCV_CHECK_ARG(Nk > 0, "attn_fwd: Nk must be > 0 (empty key set has no softmax)");
  LCV_CHECK_ARG(q_sn % 8 == 0 && k_sn % 8 == 0 && v_sn % 8 == 0 && q_sh % 8 == 0 && k_sh % 8 == 0 &&
                    v_sh % 8 == 0 && q_sb % 8 == 0 && k_sb % 8 == 0 && v_sb % 8 == 0,
                "attn_fwd: q/k/v strides must be multiples of 8 elements");
  LCV_CHECK_ARG(o_sn % 4 == 0 && o_sh % 4 == 0 && o_sb % 4 == 0, "attn_fwd: o strides must be multiples of 4 elements");
  LCV_CHECK_ARG(((uintptr_t)q % 16 == 0) && ((uintptr_t)k % 16 == 0) && ((uintptr_t)v % 16 == 0) && ((uintptr_t)o % 8 == 0),
                "attn_fwd: pointers must be 16-byte aligned");
  if (Nq == 0) return LCV_OK;
  AttnFwdParams p;
  p.q = (const bf16_t*)q; p.k = (const bf16_t*)k; p.v = (const bf16_t*)v; p.o = (bf16_t*)o; p.lse = lse;
  p.Nq = Nq; p.Nk = Nk; p.H = (int)H;
  p.q_sb = q_sb; p.q_sn = q_sn; p.q_sh = q_sh; p.k_sb = k_sb; p.k_sn = k_sn; p.k_sh = k_sh;
  p.v_sb = v_sb; p.v_sn = v_sn; p.v_sh = v_sh; p.o_sb = o_sb; p.o_sn = o_sn; p.o_sh = o_sh;
  p.scale = scale; p.scale_log2e = scale * 1.4426950408889634f;
  constexpr int NW = 8;
  const size_t lds = 2 * 2 * 64 * 256;
  const unsigned gx = (unsigned)((Nq + NW * 32 - 1) / (NW * 32));
  const char* xe = lcv_knob("LCV_ATTN_XCD");  // A/B knob: 0 disables the head-per-XCD block order
  p.gx = (int)gx;
  p.xcd_remap = ((B * H) % 8 == 0 && gx >= 8 && !(xe && xe[0] == '0')) ? 1 : 0;
  const dim3 grid = p.xcd_remap ? dim3(gx * (unsigned)(H * B)) : dim3(gx, (unsigned)H, (unsigned)B);
  auto launch = [&](auto kern) -> int {
    if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) {
      lcv_set_error("attn_fwd: cannot raise dynamic LDS");
      return LCV_EDEVICE;
    }
    hipLaunchKernelGGL(kern, grid, dim3(NW * 64), lds, (hipStream_t)stream, p);
    return LCV_OK;
  };
  // Q pre-scaled into log2 units (scale = ln 2): the multiply-free softmax body
  const bool unit = p.scale_log2e > 1.0f - 4e-7f && p.scale_log2e < 1.0f + 4e-7f;
  int rc;
  // (its LDS-DMA addresses are a scalar base + 32-bit per-lane byte offsets: one (batch, head)'s rows must span < 4 GiB)
  const bool span32 = (uint64_t)Nk * (uint64_t)(k_sn > v_sn ? k_sn : v_sn) * 2 < (1ull << 32) &&
                      (uint64_t)Nq * (uint64_t)q_sn * 2 < (1ull << 32);
  // default since round 3: 64 query rows per wave on one wave per SIMD (attn_fwd_w64.hip; equal to +4 % against the pipelined
  // two-waves-per-SIMD kernel depending on the box, profiles/r03_attn_fwd_lab.md).  A/B knob: LCV_ATTN_FWD_W64=0 = attn_fwd_pipe.hip
  const char* we = lcv_knob("LCV_ATTN_FWD_W64");
  if (unit && Nk > 512 && span32 && !(we && we[0] == '0')) {
    g_last_attn_kernel = "attn_fwd_w64_kernel";
    return attn_fwd_w64_launch(q, k, v, o, lse, B, H, Nq, Nk, q_sb, q_sn, q_sh, k_sb, k_sn, k_sh, v_sb, v_sn, v_sh, o_sb, o_sn,
                               o_sh, scale, !(xe && xe[0] == '0'), (hipStream_t)stream);
  }
  if (unit && Nk > 512 && span32) {   // LCV_ATTN_FWD_W64=0: the two-waves-per-SIMD kernel of round 2, kept as the bit-level cross-check of the default
    g_last_attn_kernel = "attn_fwd_pipe_kernel";
    return attn_fwd_pipe_launch(q, k, v, o, lse, B, H, Nq, Nk, q_sb, q_sn, q_sh, k_sb, k_sn, k_sh, v_sb, v_sn, v_sh, o_sb, o_sn,
                                o_sh, scale, !(xe && xe[0] == '0'), (hipStream_t)stream);
  }
  if (Nk <= 512) { g_last_attn_kernel = "attn_fwd_kernel<8, 0, true, 0>"; rc = launch(attn_fwd_kernel<NW, 0, true, 0>); }
  else if (unit) { g_last_attn_kernel = "attn_fwd_kernel<8, 0, false, 3>"; rc = launch(attn_fwd_kernel<NW, 0, false, 3>); }
  else { g_last_attn_kernel = "attn_fwd_kernel<8, 0, false, 1>"; rc = launch(attn_fwd_kernel<NW, 0, false, 1>); }
  if (rc != LCV_OK) return rc;
  LCV_LAUNCH_CHECK("attn_fwd");
  return LCV_OK;
}
