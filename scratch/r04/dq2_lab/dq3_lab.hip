// LAB: pass B with 32-key half tiles (generated once from attn_bwd_dq2.hip, then edited)
// Attention backward, pass B (dQ), second form - for the self-attention path where q is pre-scaled into log2 units
// (scale * log2(e) == 1; attn_fwd.hip, VAR bit 1).  Same geometry as attn_bwd_dq_kernel (8 waves x 32 query rows, 64-key tiles,
// query on the lane, S^T = K Q^T, dP^T = V dO^T, dQ^T += K^T dS^T), rebuilt like the forward kernel:
//   * K / V tiles by LDS-DMA from per-lane pointers that advance one tile per iteration (no staging registers, no ds_write);
//   * the last (possibly ragged) tile peeled, so the steady-state body has neither mask code nor ragged-row code;
//   * row constants as the initial accumulator (cdna_hip_programming.md, attention backward): the score accumulators START at
//     -lse (log2 units), so P = exp2(accumulator) needs no multiply-subtract, and the softmax scale is applied once to dQ in the
//     epilogue instead of to every dS:  dS' = P * (dP - delta)  is 2 vector instructions per score instead of 5.
// This loop, like the forward, is vector-issue bound: the file is built without SLP vectorisation (lcv_hip/build.py).
#include "lcv_common.h"
#include <stdlib.h>
#include <type_traits>
#ifndef DQ3_DEPTH
#define DQ3_DEPTH 2
#endif
#ifndef DQ3_TRD
#define DQ3_TRD 2
#endif

typedef __attribute__((address_space(3))) unsigned char lds_u8;
typedef __attribute__((address_space(1))) void gbl_void_q;
typedef __attribute__((address_space(3))) void lds_void_q;
#define AS3 __attribute__((address_space(3)))

struct AttnBwdDq2Params {
  const bf16_t* q;
  const bf16_t* k;
  const bf16_t* v;
  const bf16_t* d_o;
  const float* lse;
  const float* delta;
  bf16_t* dq;
  int64_t Nq, Nk;
  int H;
  int64_t q_sb, q_sn, q_sh, k_sb, k_sn, k_sh, v_sb, v_sn, v_sh, o_sb, o_sn, o_sh, dq_sb, dq_sn, dq_sh;
  float scale;
  int gx, xcd_remap;   // blocks per (batch, head); head-per-XCD block order (speed only: see attn_fwd.hip)
};

// <NW, NS>: waves per workgroup (32 query rows each) and K / V stages.  <8, 4>: one workgroup per CU, DMA three tiles ahead.
// <4, 2>: TWO workgroups per CU (128 KiB of LDS together) whose barriers are independent - within one workgroup the barrier
// per tile starts every wave's MFMA phase at the same moment, so one wave's exp / convert work never sits under another's
// MFMAs (the lockstep the forward kernel had before it was software-pipelined); two workgroups drift apart and overlap, at the
// price of streaming K / V into LDS twice.
template <int NW, int NS>
__global__ __launch_bounds__(NW * 64) __attribute__((amdgpu_waves_per_eu(2, 2))) void attn_bwd_dq2_kernel(const AttnBwdDq2Params p) {
  constexpr int QROWS = NW * 32;
  constexpr int NP = 16 / NW;                  // one-KiB DMA pieces per wave, tile and tensor (4 rows each)
  constexpr int TILE_BYTES = 64 * 256;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  lds_u8* lds = (lds_u8*)smem;  // [NS][K tile | V tile]

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 31, h = lane >> 5;
  // Block order (speed only): ids are dealt round-robin over the 8 XCDs, so with the remap each XCD walks the query blocks of
  // ITS OWN (batch, head) pairs and that head's K / V stream through one 4 MiB L2 instead of eight
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

  bf16x8 qf[8], dof[8];
  float sinit;   // -lse (log2 units) of this lane's query: every element of the score accumulators starts there
  float delta_q;
  {
    int64_t qrow = q0 + r;
    if (qrow > p.Nq - 1) qrow = p.Nq - 1;
    const bf16_t* qp = p.q + b * p.q_sb + qrow * p.q_sn + (int64_t)head * p.q_sh + 8 * h;
    const bf16_t* dp_ = p.d_o + b * p.o_sb + qrow * p.o_sn + (int64_t)head * p.o_sh + 8 * h;
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) {
      qf[ks] = *reinterpret_cast<const bf16x8*>(qp + 16 * ks);
      dof[ks] = *reinterpret_cast<const bf16x8*>(dp_ + 16 * ks);
    }
    sinit = -p.lse[(b * p.H + head) * p.Nq + qrow] * 1.4426950408889634f;
    delta_q = p.delta[(b * p.H + head) * p.Nq + qrow];
  }

  // LDS-DMA roles (as attn_fwd.hip): wave w fills rows 8 w .. 8 w + 7 of both tiles, 2 + 2 one-KiB instructions.  Source = a
  // scalar tile base (advanced one tile per issue by scalar adds) + a per-lane 32-bit byte offset that never changes; issued from
  // asm (lcv_common.h: a builtin DMA would be waited for before the next fragment read) and waited for at the end of the tile.
  // <4, 2> keeps ONE offset per piece (with eight the loop spilled, and a spill reload's vmcnt wait drains the DMA queue): the V
  // offset of piece i is the K offset + row_i * D, D = (v_sn - k_sn) * 2 bytes (K and V are slices of two different packed
  // tensors in the product: strides 2 H 128 and 3 H 128); row_i = (4 NP wave + 4 i) + (lane >> 4), so the first part goes into
  // the scalar base of the piece and the second is one more register.
  constexpr bool ONE_OFF = NW == 4;
  unsigned koff[NP], voff[ONE_OFF ? 1 : NP];
  const int64_t kv_delta = (p.v_sn - p.k_sn) * 2;
  auto lane16 = []() -> int { return (int)(__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)) >> 4); };
  auto dma_row_of = [&](int i) { return 4 * NP * wave + 4 * i + lane16(); };
  // recomputed from the hardware lane id at every use: a register held across the loop for it tipped three offsets into scratch
  // (wraps for a negative delta; the SUM with koff is a true offset >= 0)
  auto vlane = [&]() -> unsigned { return (unsigned)lane16() * (unsigned)kv_delta; };
#pragma unroll
  for (int i = 0; i < NP; ++i) {
    const int row = 4 * NP * wave + 4 * i + (lane >> 4);
    const int col = 8 * ((lane & 15) ^ (((row & 3) << 2) | ((row >> 2) & 3)));
    koff[i] = (unsigned)((row * p.k_sn + col) * 2);
    if constexpr (!ONE_OFF) voff[i] = (unsigned)((row * p.v_sn + col) * 2);
  }
  const char* kbase_u = lcv_uniform_ptr(kbase);
  const char* vbase_u = lcv_uniform_ptr(vbase);
  const unsigned lds_addr0 = (unsigned)(uintptr_t)lds;
  auto dma_tile = [&](int t, int buf, auto full_c) {
    constexpr bool FULL = decltype(full_c)::value;
    const unsigned dst = lds_addr0 + (unsigned)(buf * 2 * TILE_BYTES) + (unsigned)wave * (unsigned)(NP * 1024);
    const char* kt = kbase_u + (int64_t)t * (128 * p.k_sn);   // scalar: 64 rows x stride x 2 bytes per tile
    const char* vt = vbase_u + (int64_t)t * (128 * p.v_sn);
    if (FULL || (int64_t)t * 64 + 64 <= p.Nk) {
#pragma unroll
      for (int i = 0; i < NP; ++i) {
        lcv_lds_dma16_sv(koff[i], kt, dst + 1024u * i);
        if constexpr (ONE_OFF) lcv_lds_dma16_sv(koff[i] + vlane(), vt + (4 * NP * wave + 4 * i) * kv_delta, dst + (unsigned)TILE_BYTES + 1024u * i);
        else lcv_lds_dma16_sv(voff[i], vt, dst + (unsigned)TILE_BYTES + 1024u * i);
      }
    } else {  // ragged last tile (once per workgroup): rows past Nk re-read the last key (masked below)
#pragma unroll
      for (int i = 0; i < NP; ++i) {
        int64_t back = (int64_t)t * 64 + dma_row_of(i) - (p.Nk - 1);
        if (back < 0) back = 0;
        lcv_lds_dma16(kt + koff[i] - back * p.k_sn * 2, dst + 1024u * i);
        const int64_t vo = ONE_OFF ? (int64_t)koff[i] + (int64_t)dma_row_of(i) * kv_delta : (int64_t)voff[ONE_OFF ? 0 : i];
        lcv_lds_dma16(vt + vo - back * p.v_sn * 2, dst + (unsigned)TILE_BYTES + 1024u * i);
      }
    }
  };

  const int kfz = ((r & 3) << 2) | ((r >> 2) & 3);
  const int k_row_off = 256 * r;
  const int q4 = (lane >> 2) & 3, p4 = lane & 3, g1 = (lane >> 4) & 1;
  int t_base[2], t_low[2];
#pragma unroll
  for (int half = 0; half < 2; ++half) {
    t_base[half] = 256 * (4 * h + 8 * half + q4) + 8 * (p4 & 1);
    t_low[half] = (2 * g1 + (p4 >> 1)) ^ (h + 2 * half);
  }

  f32x16 dqacc[4];
#pragma unroll
  for (int d = 0; d < 4; ++d)
#pragma unroll
    for (int e = 0; e < 16; ++e) dqacc[d][e] = 0.f;

  // K / V stages: a ring of NS = 4 (128 KiB; one workgroup per CU anyway: ~220 registers per wave), the DMA runs three tiles
  // ahead.  With two stages a tile had one tile's worth of MFMAs (~1.6 us) to land and the waves still waited 45 % of their
  // cycles; counted waits: at the end of tile t only tile t+1 has to be in, t+2 and t+3 stay in flight (4 pieces each).
  static_assert((NW == 8 && NS == 4) || (NW == 4 && NS == 2), "the counted waits below are written for these two forms");
  const int nt = (int)((p.Nk + 63) / 64);
#pragma unroll
  for (int i = 0; i < NS - 1; ++i)
    if (i < nt) dma_tile(i, i, std::false_type{});
  if (NS == 4 && nt >= 3) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
  else if (NS == 4 && nt == 2) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
  else __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0), visible to hipcc's wait insertion (an asm wait is not): no conservative vmcnt(N) for the row constants inside the loop
  __syncthreads();

  auto tile_body = [&](const int t, auto last_c, auto full_dma_c) {
    constexpr bool has_next = !decltype(last_c)::value;
    const int buf = t & (NS - 1);
    if (decltype(full_dma_c)::value || t + NS - 1 < nt) dma_tile(t + NS - 1, (t + NS - 1) & (NS - 1), full_dma_c);   // the stage tile t-1 has just left
    const lds_u8* kb = lds + buf * 2 * TILE_BYTES;
    const lds_u8* vb = kb + TILE_BYTES;

    // two 32-key halves, one after the other: 32 accumulator registers for S / dP instead of 64, and what that frees holds K / V
    // fragments TWO steps ahead of the MFMAs that use them (at 256 registers hipcc issued every fragment read directly in front
    // of its MFMA: 29 full LDS latencies per tile on the issue path)
#pragma unroll
    for (int half = 0; half < 2; ++half) {
      const lds_u8* kh = kb + half * 32 * 256 + k_row_off;
      const lds_u8* vh = vb + half * 32 * 256 + k_row_off;
      auto co = [&](int ks) { return 16 * ((2 * ks + h) ^ kfz); };
      f32x16 s, d;
#pragma unroll
      for (int e = 0; e < 16; ++e) { s[e] = sinit; d[e] = 0.f; }
      bf16x8 af[DQ3_DEPTH + 1], cf[DQ3_DEPTH + 1];
#pragma unroll
      for (int i = 0; i < DQ3_DEPTH; ++i) {
        af[i] = *reinterpret_cast<const AS3 bf16x8*>(kh + co(i));
        cf[i] = *reinterpret_cast<const AS3 bf16x8*>(vh + co(i));
      }
#pragma unroll
      for (int ks = 0; ks < 8; ++ks) {
        if (ks + DQ3_DEPTH < 8) {
          af[(ks + DQ3_DEPTH) % (DQ3_DEPTH + 1)] = *reinterpret_cast<const AS3 bf16x8*>(kh + co(ks + DQ3_DEPTH));
          cf[(ks + DQ3_DEPTH) % (DQ3_DEPTH + 1)] = *reinterpret_cast<const AS3 bf16x8*>(vh + co(ks + DQ3_DEPTH));
        }
        s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[ks % (DQ3_DEPTH + 1)], qf[ks], s, 0, 0, 0);
        d = __builtin_amdgcn_mfma_f32_32x32x16_bf16(cf[ks % (DQ3_DEPTH + 1)], dof[ks], d, 0, 0, 0);
#ifdef DQ3_SGB
        __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);   // 2 DS reads
        __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);   // 2 MFMA
#endif
      }
      if (!has_next && (p.Nk & 63)) {  // last tile only
        const int valid = (int)(p.Nk - (int64_t)t * 64) - 32 * half;
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const int key = (e & 3) + 8 * (e >> 2) + 4 * h;
          if (key >= valid) s[e] = -INFINITY;
        }
      }
      // K^T fragments of this half's 8 dQ products (i = 2 kk2 + ... : kk2 = i >> 2, dd = i & 3), DQ3_TRD products ahead; the
      // first ones are requested BEFORE the vector work on the scores, which hides their latency
      auto ktf_read = [&](int i) {
        const int kk = 2 * half + (i >> 2), dd = i & 3;
        const int dx = 64 * (dd ^ q4);
        const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((AS3 s16x4*)(kb + t_base[0] + 4096 * kk + dx + 16 * t_low[0]));
        const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((AS3 s16x4*)(kb + t_base[1] + 4096 * kk + dx + 16 * t_low[1]));
        return __builtin_bit_cast(bf16x8, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
      };
      bf16x8 ktf[DQ3_TRD + 1];
#pragma unroll
      for (int i = 0; i < DQ3_TRD; ++i) ktf[i] = ktf_read(i);
#pragma unroll
      for (int e = 0; e < 16; ++e) s[e] = __builtin_amdgcn_exp2f(s[e]) * (d[e] - delta_q);
      bf16x8 dsb[2];
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        dsb[0][j] = (__bf16)s[j];
        dsb[1][j] = (__bf16)s[8 + j];
      }
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        if (i + DQ3_TRD < 8) ktf[(i + DQ3_TRD) % (DQ3_TRD + 1)] = ktf_read(i + DQ3_TRD);
        dqacc[i & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ktf[i % (DQ3_TRD + 1)], dsb[i >> 2], dqacc[i & 3], 0, 0, 0);
      }
    }
    if (has_next) {   // tile t+1 is in; what was requested after it may still be in flight
      const int later = nt - 2 - t;   // tiles requested after t+1 (at most NS - 2 of them are outstanding)
      if (NS == 4 && later >= 2) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
      else if (NS == 4 && later == 1) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
      else lcv_dma_wait_all();
    }
    __syncthreads();
  };
  // steady state: the tile requested (t + NS - 1) is a whole one - no ragged-row code, and none of its values, in the loop
  int t = 0;
  for (; t + NS - 1 < nt - 1; ++t) tile_body(t, std::false_type{}, std::true_type{});
  for (; t + 1 < nt; ++t) tile_body(t, std::false_type{}, std::false_type{});
  tile_body(nt - 1, std::true_type{}, std::false_type{});   // the (possibly ragged) last tile: the only body with mask code

  const int64_t qrow = q0 + r;
  if (qrow < p.Nq) {
    bf16_t* dqp = p.dq + b * p.dq_sb + qrow * p.dq_sn + (int64_t)head * p.dq_sh;
#pragma unroll
    for (int d = 0; d < 4; ++d)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        u16x4 pk;
#pragma unroll
        for (int e = 0; e < 4; ++e) pk[e] = f2bf(dqacc[d][4 * i + e] * p.scale);
        *reinterpret_cast<u16x4*>(dqp + 32 * d + 8 * i + 4 * h) = pk;
      }
  }
}

// called by lcv_attn_bwd (attn_bwd.hip) when scale * log2(e) == 1 and LCV_ATTN_BWD_VAR != 0
int attn_bwd_dq2_launch(const void* q, const void* k, const void* v, const void* d_o, const float* lse, const float* delta,
                        void* dq, int64_t B, int64_t H, int64_t Nq, int64_t Nk, int64_t q_sb, int64_t q_sn, int64_t q_sh,
                        int64_t k_sb, int64_t k_sn, int64_t k_sh, int64_t v_sb, int64_t v_sn, int64_t v_sh, int64_t o_sb,
                        int64_t o_sn, int64_t o_sh, int64_t dq_sb, int64_t dq_sn, int64_t dq_sh, float scale, hipStream_t s) {
  AttnBwdDq2Params p;
  p.q = (const bf16_t*)q; p.k = (const bf16_t*)k; p.v = (const bf16_t*)v; p.d_o = (const bf16_t*)d_o;
  p.lse = lse; p.delta = delta; p.dq = (bf16_t*)dq; p.Nq = Nq; p.Nk = Nk; p.H = (int)H;
  p.q_sb = q_sb; p.q_sn = q_sn; p.q_sh = q_sh; p.k_sb = k_sb; p.k_sn = k_sn; p.k_sh = k_sh;
  p.v_sb = v_sb; p.v_sn = v_sn; p.v_sh = v_sh; p.o_sb = o_sb; p.o_sn = o_sn; p.o_sh = o_sh;
  p.dq_sb = dq_sb; p.dq_sn = dq_sn; p.dq_sh = dq_sh; p.scale = scale;
  const char* we = lcv_knob("LCV_ATTN_BWD_DQ_WAVES");   // A/B knob: 8 = one 8-wave workgroup per CU (4 stages), 4 = two 4-wave ones
  const int nw = (we && we[0] == '8') ? 8 : 4;
  const size_t lds = (nw == 8 ? 4 : 2) * 2 * 64 * 256;   // NS stages of (K tile | V tile)
  // (function-local static: initialised once, thread-safe)
  static const bool attr_ok = !(hipFuncSetAttribute((const void*)attn_bwd_dq2_kernel<8, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, 4 * 2 * 64 * 256) != hipSuccess ||
        hipFuncSetAttribute((const void*)attn_bwd_dq2_kernel<4, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * 2 * 64 * 256) != hipSuccess);
  if (!attr_ok) {
      lcv_set_error("attn_bwd: cannot raise dynamic LDS");
      return LCV_EDEVICE;
  }
  const unsigned gx = (unsigned)((Nq + nw * 32 - 1) / (nw * 32));
  // A/B knob LCV_ATTN_BWD_XCD=1 enables the head-per-XCD block order.  OFF by default: at the K3-TTA shapes (25 200 keys x 32
  // heads) it measured 27.06 vs 26.51 ms per layer in one process - unlike the forward, these passes are not helped by it
  const char* xe = lcv_knob("LCV_ATTN_BWD_XCD");
  p.gx = (int)gx;
  p.xcd_remap = ((B * H) % 8 == 0 && gx >= 8 && xe && xe[0] == '1') ? 1 : 0;
  const dim3 grid = p.xcd_remap ? dim3(gx * (unsigned)(H * B)) : dim3(gx, (unsigned)H, (unsigned)B);
  if (nw == 8) hipLaunchKernelGGL((attn_bwd_dq2_kernel<8, 4>), grid, dim3(512), lds, s, p);
  else hipLaunchKernelGGL((attn_bwd_dq2_kernel<4, 2>), grid, dim3(256), lds, s, p);
  LCV_LAUNCH_CHECK("attn_bwd_dq2");
  return LCV_OK;
}
