// Attention backward, pass A (dK, dV), third form (round 3) - for the self-attention path where q is pre-scaled into log2
// units (scale * log2(e) == 1).  Same arithmetic, operand maps, LDS tile image and rounding points as attn_bwd_dkv2_kernel
// (bit-identical results); what changes is the ORDER of each wave's instruction stream:
//
//   * workgroup = 4 waves = 128 keys, ONE wave per SIMD (dkv2: two 4-wave workgroups per CU whose waves hide each other's
//     stalls by luck of their phase).  K and V rows of the wave's 32 keys stay in registers as B operands; dK^T / dV^T are 128
//     accumulator registers, pinned to AGPRs;
//   * software pipeline inside the wave (the transformation that took the forward from 0.60 to 0.67 of the matrix pipe,
//     attn_fwd_pipe.hip): iteration i runs
//         phase Y   dV += dO(i-1)^T P(i-1), dK += Q(i-1)^T dS(i-1)   16 MFMAs || P(i) = exp2(S(i)), dS(i) = P(i) dP(i), bf16
//                                                                      packing (48 vector instructions) and the LDS-DMA
//                                                                      requests of tile i+2, all in the MFMA gaps
//         phase X   S(i+1) = Q K^T - lse, dP(i+1) = dO V^T - delta   16 MFMAs
//         barrier   (one per tile)
//     so no MFMA waits for an exponential, and the first fragments of every phase are requested in the last gaps of the phase
//     before it (the transposed reads of Y before the barrier: their tile landed two barriers ago);
//   * Q / dO tiles on a ring of SIX stages: tile i-1 is read transposed in Y while tile i+1 is read by rows in X, tile i is
//     kept for the next Y, tiles i+2 .. i+4 have landed or are in flight.  An LDS-DMA lands ~1 us (~2 000 cycles) after its
//     issue and a tile now takes ~1 100: a request made one tile ahead (dkv2) would be waited for; made three tiles ahead it is
//     not.  The wait at the end of an iteration is COUNTED: the two newest tiles stay in flight across the barrier.
// The vector instructions of the gaps and the MFMAs are asm volatile and fenced (sched_barrier): the order in this file is the
// order in the binary.
#include "lcv_common.h"
#include <stdlib.h>
#include <type_traits>

typedef __attribute__((address_space(3))) unsigned char lds_u8x;
#define AS3X __attribute__((address_space(3)))
#define SCHED_FENCE() __builtin_amdgcn_sched_barrier(0)

struct AttnBwdDkv3Params {
  const bf16_t* q;
  const bf16_t* k;
  const bf16_t* v;
  const bf16_t* d_o;
  const float* consts;   // per (b, h): nlse2[Nqp] | ndelta[Nqp] (attn_bwd_delta_kernel), Nqp = roundup(Nq, 32)
  bf16_t* dk;
  bf16_t* dv;
  int64_t Nq, Nk;
  int H;
  int64_t q_sb, q_sn, q_sh, k_sb, k_sn, k_sh, v_sb, v_sn, v_sh, o_sb, o_sn, o_sh;
  int64_t dk_sb, dk_sn, dk_sh, dv_sb, dv_sn, dv_sh;
  float scale;
  int accumulate_kv;
};

__device__ __forceinline__ int swz_k3(int row) { return ((row & 3) << 2) | ((row >> 2) & 3); }
__device__ __forceinline__ float x_exp2(float x) { float y; asm volatile("v_exp_f32 %0, %1" : "=v"(y) : "v"(x)); return y; }
__device__ __forceinline__ float x_mul(float a, float b) { float y; asm volatile("v_mul_f32 %0, %1, %2" : "=v"(y) : "v"(a), "v"(b)); return y; }
__device__ __forceinline__ unsigned x_pack(float lo, float hi) { unsigned y; asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(y) : "v"(lo), "v"(hi)); return y; }

// MFMAs from asm with the register CLASS of the accumulator pinned: left to itself hipcc puts every MFMA result of a kernel
// with more than 256 registers into AGPRs - the score tiles too, which then cost a v_accvgpr_read per element before the
// exponentials.  Score tiles live in arch VGPRs ("v"), dK / dV accumulators in AGPRs ("a").
// (asm is opaque to the hazard recogniser: every result below is read >= 3 MFMA slots after the MFMA that wrote it.)
// the same with the B operand in AGPRs: the K / V rows of the wave's keys are loaded once and only ever read by MFMAs, so they
// sit in the half of the register file the vector instructions cannot address (64 arch VGPRs back for the second score set)
// GUARD: outside the straight-line steady loop hipcc re-homes accumulators and operands between code paths with v_accvgpr_mov /
// v_mov copies placed DIRECTLY in front of the asm that reads them; the recogniser cannot see the MFMA inside, so the wait states
// of "VALU write -> MFMA read" are spelled out (4 of them; measured symptom without: dK of small shapes wrong and not repeatable)
template <bool GUARD = true>
__device__ __forceinline__ void mfma_vb(f32x16& c, const bf16x8& a, const bf16x8& b) {
  if constexpr (GUARD) asm volatile("s_nop 3\n\tv_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(c) : "v"(a), "a"(b));
  else asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(c) : "v"(a), "a"(b));
}
template <bool GUARD = true>
__device__ __forceinline__ void mfma_a(f32x16& c, const bf16x8& a, const bf16x8& b) {
  if constexpr (GUARD) asm volatile("s_nop 3\n\tv_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(c) : "v"(a), "v"(b));
  else asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(c) : "v"(a), "v"(b));
}

// Diagnostic build only (scratch/attn_lab/build_dkv3.sh defines LCV_DKV3_STAMPS; the product never does): s_memtime stamps of
// waves 0 and 2 of one workgroup at five points of eight consecutive iterations, parked in LDS behind the stages.
#ifdef LCV_DKV3_STAMPS
__device__ unsigned long long* g_dkv3_dbg = nullptr;
__device__ int g_dkv3_dbg_block = 0;
#define DKV3_STAMP(it, id)                                                                                  \
  if (dbg_on && (it) >= 100 && (it) < 108) {                                                                \
    unsigned long long t_;                                                                                  \
    __builtin_amdgcn_sched_barrier(0);                                                                      \
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");                              \
    __builtin_amdgcn_sched_barrier(0);                                                                      \
    if (lane_now() == 0) *reinterpret_cast<AS3X unsigned long long*>(lds + 7 * STAGE + wave * 1024 + (((it) - 100) * 8 + (id)) * 8) = t_; \
  }
extern "C" void attn_dkv3_set_stamps(unsigned long long* buf, int block) {
  (void)hipMemcpyToSymbol(HIP_SYMBOL(g_dkv3_dbg), &buf, sizeof(buf));
  (void)hipMemcpyToSymbol(HIP_SYMBOL(g_dkv3_dbg_block), &block, sizeof(block));
}
#else
#define DKV3_STAMP(it, id)
#endif

// packed bf16 P / dS of one 32-query tile: word [ss][j] = elements (8 ss + 2 j, 8 ss + 2 j + 1) of the lane's 16
struct PackedTile {
  unsigned p[2][4];
  unsigned ds[2][4];
};

__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void attn_bwd_dkv3_kernel(const AttnBwdDkv3Params p) {
  constexpr int QT = 32;
  constexpr int TILE_BYTES = QT * 256;                  // one [32][128] bf16 tile
  constexpr int STAGE = 2 * TILE_BYTES + 2 * QT * 4;    // Q | dO | -lse (log2 units) | -delta
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  constexpr int NST = 7;
  lds_u8x* lds = (lds_u8x*)smem;                        // [NST] stages

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 31, h = lane >> 5;
  const int kb = blockIdx.x, head = blockIdx.y;
  const int64_t b = blockIdx.z;
  const int64_t key0 = (int64_t)kb * 128;

  // ---- K and V rows of this lane's key as B operands (registers) ----
  bf16x8 kf[8], vf[8];
  {
    int64_t krow = key0 + wave * 32 + r;
    if (krow > p.Nk - 1) krow = p.Nk - 1;
    const bf16_t* kp = p.k + b * p.k_sb + krow * p.k_sn + (int64_t)head * p.k_sh + 8 * h;
    const bf16_t* vp = p.v + b * p.v_sb + krow * p.v_sn + (int64_t)head * p.v_sh + 8 * h;
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) {
      kf[ks] = *reinterpret_cast<const bf16x8*>(kp + 16 * ks);
      vf[ks] = *reinterpret_cast<const bf16x8*>(vp + 16 * ks);
    }
    // into AGPRs HERE, long before the first MFMA that reads them: left to hipcc, the v_accvgpr_write copies land right in front
    // of the first asm MFMA, which (opaque to the hazard recogniser) then reads them without the required wait states
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) asm volatile("" : "+a"(kf[ks]), "+a"(vf[ks]));
  }

  // ---- Q / dO tile staging by asm-issued LDS-DMA (2 + 2 pieces per wave), the row constants by one dword piece (wave 0) ----
  const bf16_t* qbase = p.q + b * p.q_sb + (int64_t)head * p.q_sh;
  const bf16_t* dobase = p.d_o + b * p.o_sb + (int64_t)head * p.o_sh;
  const int64_t Nqp = (p.Nq + 31) / 32 * 32;
  const char* cbase_u = lcv_uniform_ptr(p.consts + (b * p.H + head) * 2 * Nqp);
  const unsigned coff = (unsigned)((lane < 32 ? lane : Nqp + lane - 32) * 4);   // one dword piece: 32 x nlse2 | 32 x ndelta
  unsigned qoff[2], dooff[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int row = 4 * (2 * wave + i) + (lane >> 4);
    const int col = 8 * ((lane & 15) ^ swz_k3(row));
    qoff[i] = (unsigned)((row * p.q_sn + col) * 2);
    dooff[i] = (unsigned)((row * p.o_sn + col) * 2);
  }
  const char* qbase_u = lcv_uniform_ptr(qbase);
  const char* dobase_u = lcv_uniform_ptr(dobase);
  const unsigned stage_addr0 = (unsigned)(uintptr_t)lds;
  auto lane_now = []() -> int { return (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)); };
#ifdef LCV_DKV3_STAMPS
  const bool dbg_on = g_dkv3_dbg != nullptr && (int)blockIdx.x == g_dkv3_dbg_block && blockIdx.y == 0 && (wave == 0 || wave == 2);
#endif
  // piece `which` (0, 1: Q rows; 2, 3: dO rows; 4: the row constants) of the tile that starts at query q0 -> stage at byte stage_off
  auto load_piece = [&](int which, int64_t q0, int stage_off, bool known_full = false) {
    const unsigned sb = stage_addr0 + (unsigned)stage_off + (unsigned)wave * 2048u;
    if (which == 4) {   // EVERY wave requests the 256 bytes (same bytes to the same place): five requests per tile and wave, one wait count
      lcv_lds_dma4_sv(coff, cbase_u + q0 * 4, stage_addr0 + (unsigned)(stage_off + 2 * TILE_BYTES));
      return;
    }
    const int i = which & 1;
    const bool is_do = which >= 2;
    const char* base = (is_do ? dobase_u : qbase_u) + q0 * (2 * (is_do ? p.o_sn : p.q_sn));
    const unsigned off = is_do ? dooff[i] : qoff[i];
    const unsigned dst = sb + (is_do ? (unsigned)TILE_BYTES : 0u) + 1024u * i;
    if (known_full || q0 + QT <= p.Nq) {
      lcv_lds_dma16_sv(off, base, dst);
    } else {                                             // ragged last tile (once per workgroup): rows past Nq re-read the last query
      int64_t back = q0 + 4 * (2 * wave + i) + (lane_now() >> 4) - (p.Nq - 1);
      if (back < 0) back = 0;
      lcv_lds_dma16(base + off - back * (is_do ? p.o_sn : p.q_sn) * 2, dst);
    }
  };
  auto load_tile = [&](int64_t q0, int stage_off) {
#pragma unroll
    for (int w = 0; w < 5; ++w) load_piece(w, q0, stage_off);
  };

  // ---- per-lane LDS read offsets ----
  const int rf = swz_k3(r);
  int row_off[8];                                        // row r of a [32][128] tile, k-step ks: bytes
#pragma unroll
  for (int ks = 0; ks < 8; ++ks) row_off[ks] = 256 * r + 16 * ((2 * ks + h) ^ rf);
  const int q4 = (lane >> 2) & 3, p4 = lane & 3, g1 = (lane >> 4) & 1;
  int t_off[2][4];                                       // transposed reads: [half][d]
#pragma unroll
  for (int half = 0; half < 2; ++half)
#pragma unroll
    for (int d = 0; d < 4; ++d)
      t_off[half][d] = 256 * (4 * h + 8 * half + q4) + 8 * (p4 & 1) + 64 * (d ^ q4) + 16 * ((2 * g1 + (p4 >> 1)) ^ (h + 2 * half));

  f32x16 dkacc[4], dvacc[4];
#pragma unroll
  for (int d = 0; d < 4; ++d)
#pragma unroll
    for (int e = 0; e < 16; ++e) { dkacc[d][e] = 0.f; dvacc[d][e] = 0.f; }

  const int nt = (int)((p.Nq + QT - 1) / QT);
  // ---- prologue: tiles 0 .. 5 requested; everything landed before the first read ----
#pragma unroll
  for (int i = 0; i < 6; ++i)
    if (i < nt) load_tile((int64_t)i * QT, i * STAGE);
  lcv_dma_wait_all();
  __syncthreads();

  // two score sets: while the vector work of tile i reads one, the S / dP chains of tile i + 1 accumulate into the other
  f32x16 sA, dA, sB, dB;
  // fragment rings: an LDS read returns well over two MFMA slots after its issue, so fragments are requested PD steps ahead
  constexpr int PD = 2, RING = PD + 1;
  bf16x8 aq[RING], ad[RING];                             // row fragments (S / dP chains)
  s16x4 dlo[RING], dhi[RING], qlo[RING], qhi[RING];      // transposed fragments (dV / dK products)

  // requests: the row constants of the tile in stage `sx` into a score set (the initial accumulators: element e <-> query
  // (e & 3) + 8 (e >> 2) + 4 h), its row fragments of k-step ks, the transposed fragments of group g = 4 ss + d of the tile in `sy`
  auto ld_consts = [&](int sx, f32x16& S, f32x16& D) {
    const lds_u8x* lb = lds + sx + 2 * TILE_BYTES;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const f32x4 l4 = *reinterpret_cast<const AS3X f32x4*>(lb + (8 * g + 4 * h) * 4);
      const f32x4 d4 = *reinterpret_cast<const AS3X f32x4*>(lb + QT * 4 + (8 * g + 4 * h) * 4);
#pragma unroll
      for (int e = 0; e < 4; ++e) { S[4 * g + e] = l4[e]; D[4 * g + e] = d4[e]; }
    }
  };
  auto ld_x = [&](int sx, int ks, int st) {
    aq[st] = *reinterpret_cast<const AS3X bf16x8*>(lds + sx + row_off[ks]);
    ad[st] = *reinterpret_cast<const AS3X bf16x8*>(lds + sx + TILE_BYTES + row_off[ks]);
  };
  auto ld_y = [&](int sy, int g, int st) {
    const int ss = g >> 2, d = g & 3;
    dlo[st] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((AS3X s16x4*)(lds + sy + TILE_BYTES + 4096 * ss + t_off[0][d]));
    dhi[st] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((AS3X s16x4*)(lds + sy + TILE_BYTES + 4096 * ss + t_off[1][d]));
    qlo[st] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((AS3X s16x4*)(lds + sy + 4096 * ss + t_off[0][d]));
    qhi[st] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((AS3X s16x4*)(lds + sy + 4096 * ss + t_off[1][d]));
  };

  // the vector work of one tile: P = exp2(S), dS' = P * dP (the scale goes to dK in the epilogue), packs.  `step(i)` issues the
  // exponential of element i, the product of element i - 1 and the packs of the pairs that have just completed; results go to
  // short-lived scalars, never back into the score tuples.  18 steps (0 .. 17) per tile.
  float pe[16], de[16];
  auto valu_step = [&](int i, PackedTile& out, const f32x16& S, const f32x16& D) {
    if (i < 16) pe[i] = x_exp2(S[i]);
    if (i >= 1 && i <= 16) {
      const int j = i - 1;
      de[j] = x_mul(pe[j], D[j]);
      if (j & 1) out.p[j >> 3][(j & 7) >> 1] = x_pack(pe[j - 1], pe[j]);
    }
    if (i >= 2 && i <= 17) {
      const int j = i - 2;
      if (j & 1) out.ds[j >> 3][(j & 7) >> 1] = x_pack(de[j - 1], de[j]);
    }
  };
  auto mfma_y = [&](int g, int m, const PackedTile& cur, auto guard_c) {   // product m (0: dV, 1: dK) of group g = 4 ss + d
    constexpr bool GUARD = decltype(guard_c)::value;
    const int ss = g >> 2, d = g & 3, st = g % RING;
    if (m) {
      const bf16x8 qtf = __builtin_bit_cast(bf16x8, __builtin_shufflevector(qlo[st], qhi[st], 0, 1, 2, 3, 4, 5, 6, 7));
      const u32x4 wd = {cur.ds[ss][0], cur.ds[ss][1], cur.ds[ss][2], cur.ds[ss][3]};
      mfma_a<GUARD>(dkacc[d], qtf, __builtin_bit_cast(bf16x8, wd));
    } else {
      const bf16x8 dof = __builtin_bit_cast(bf16x8, __builtin_shufflevector(dlo[st], dhi[st], 0, 1, 2, 3, 4, 5, 6, 7));
      const u32x4 wp = {cur.p[ss][0], cur.p[ss][1], cur.p[ss][2], cur.p[ss][3]};
      mfma_a<GUARD>(dvacc[d], dof, __builtin_bit_cast(bf16x8, wp));
    }
  };

  // plain phases (pipeline prologue and the last tile): S / dP of the tile in stage `sx` into (S, D), constants and the first PD
  // k-steps requested by the caller; products of the tile in stage `sy` with the packs `cur`, first PD groups requested by the caller
  auto phase_x = [&](int sx, f32x16& S, f32x16& D) {
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) {
      if (ks + PD < 8) ld_x(sx, ks + PD, (ks + PD) % RING);
      mfma_vb(S, aq[ks % RING], kf[ks]);
      SCHED_FENCE();
      mfma_vb(D, ad[ks % RING], vf[ks]);
      SCHED_FENCE();
    }
  };
  auto phase_y = [&](int sy, const PackedTile& cur) {
#pragma unroll
    for (int g = 0; g < 8; ++g) {
      if (g + PD < 8) ld_y(sy, g + PD, (g + PD) % RING);
      mfma_y(g, 0, cur, std::true_type{});
      SCHED_FENCE();
      mfma_y(g, 1, cur, std::true_type{});
      SCHED_FENCE();
    }
    // the two tails (odd / even tile count) end with the accumulators in different registers and hipcc copies one set over -
    // v_accvgpr_mov straight behind the last (opaque) MFMAs read results that are not back yet: d-blocks of dK came out without
    // their last group.  The copies now sit behind this statement's wait states.
    asm volatile("s_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15"
                 : "+a"(dkacc[0]), "+a"(dkacc[1]), "+a"(dkacc[2]), "+a"(dkacc[3]), "+a"(dvacc[0]), "+a"(dvacc[1]), "+a"(dvacc[2]), "+a"(dvacc[3]));
  };

  // stage byte offsets of tiles i - 1, i, i + 1, i + 2 and i + 5 at the top of iteration i: tile t lives in stage t % NST
  int s_m1 = 0, s_0 = STAGE, s_p1 = 2 * STAGE, s_p2 = 3 * STAGE, s_p5 = 6 * STAGE;
  auto adv = [&](int x) { return x + STAGE == NST * STAGE ? 0 : x + STAGE; };

  // ONE merged phase per tile: the MFMAs alternate between the S / dP chains of tile i + 1 and the products of tile i - 1,
  //     S(ks = g)   dV(g)   dP(ks = g)   dK(g)        g = 0 .. 7,
  // so a chain's next link is four MFMA slots away (with one wave per SIMD and two chains back to back, a link waited for its
  // predecessor: 47 cycles per MFMA measured) and the 54 vector instructions of tile i, the 5 LDS-DMA requests of tile i + 5 and
  // the fragment reads spread over 32 gaps.  Preconditions (left by the previous iteration): (AS, AD) hold the row constants of tile
  // i + 1, k-steps 0 .. PD - 1 of tile i + 1 and groups 0 .. PD - 1 of tile i - 1 are requested.  (RS, RD) = S / dP of tile i.
  auto merged = [&](int i, const PackedTile& cur, PackedTile& nxt, f32x16& RS, f32x16& RD, f32x16& AS, f32x16& AD, auto steady_c) {
    constexpr bool STEADY = decltype(steady_c)::value;     // tiles up to i + 5 exist and are full: straight-line code
    const bool has_x = STEADY || i + 1 < nt;               // tile i + 1 exists: its chains run
    const bool has_x2 = STEADY || i + 2 < nt;              // tile i + 2 exists: its constants / first fragments are requested at the end
    const bool has_dma = STEADY || i + 5 < nt;
    DKV3_STAMP(i, 0)
    // the last tile's iteration has no S / dP MFMAs in front of its vector work: the dP chain of tile i ended two MFMA slots ago
    // and its results are not back yet (symptom: dK of the last tile wrong, the same wrong every run).  Once per workgroup.
    if constexpr (!STEADY)
      if (!has_x) asm volatile("s_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15" ::: "memory");
#pragma unroll
    for (int g = 0; g < 8; ++g) {
      if (g + PD < 8) {
#ifndef LCV_DKV3_NO_ROW
        if (has_x) ld_x(s_p1, g + PD, (g + PD) % RING);
#endif
#ifndef LCV_DKV3_NO_TR
        ld_y(s_m1, g + PD, (g + PD) % RING);
#endif
      }
#pragma unroll
      for (int m = 0; m < 4; ++m) {
        // unguarded only where the code is straight-line and the copies (if any) sit at the top of the iteration: g >= 1 of STEADY
        if (g >= 1 && STEADY) {
          if (m == 0) mfma_vb<false>(AS, aq[g % RING], kf[g]);
          else if (m == 2) mfma_vb<false>(AD, ad[g % RING], vf[g]);
          else mfma_y(g, m >> 1, cur, std::false_type{});
        } else {
          if (m == 0) { if (has_x) mfma_vb<true>(AS, aq[g % RING], kf[g]); }
          else if (m == 2) { if (has_x) mfma_vb<true>(AD, ad[g % RING], vf[g]); }
          else mfma_y(g, m >> 1, cur, std::true_type{});
        }
        SCHED_FENCE();
        const int gap = 4 * g + m;                        // 0 .. 31
        // 18 vector steps at gaps 0, 1, 3, 4, 6, 7, 9, 10, 12, 14, 15, 17, 18, 20, 21, 23, 24, 26 (= floor(28 j / 18))
#pragma unroll
        for (int j = 0; j < 18; ++j)
#ifndef LCV_DKV3_NO_VALU
          if ((28 * j) / 18 == gap) valu_step(j, nxt, RS, RD);
#else
          if ((28 * j) / 18 == gap && j == 0) valu_step(j, nxt, RS, RD);
#endif
        // the five LDS-DMA requests of tile i + 5 in gaps that carry no vector step
#ifdef LCV_DKV3_NO_DMA
        if (false) {
#else
        if (has_dma) {
#endif
          if (gap == 2) load_piece(0, (int64_t)(i + 5) * QT, s_p5, STEADY);
          if (gap == 5) load_piece(1, (int64_t)(i + 5) * QT, s_p5, STEADY);
          if (gap == 8) load_piece(2, (int64_t)(i + 5) * QT, s_p5, STEADY);
          if (gap == 11) load_piece(3, (int64_t)(i + 5) * QT, s_p5, STEADY);
          if (gap == 13) load_piece(4, (int64_t)(i + 5) * QT, s_p5, STEADY);
        }
        // what the next iteration expects: the row constants of tile i + 2 (into the set the vector work has just left: its last
        // read was in gap 26), its first k-steps, and the first groups of tile i (all landed before the previous barrier)
#if !defined(LCV_DKV3_NO_ROW) && !defined(LCV_DKV3_NO_TR)
        if (gap == 28 && has_x2) ld_consts(s_p2, RS, RD);
        if (gap == 29) { if (has_x2) ld_x(s_p2, 0, 0); ld_y(s_0, 0, 0); }
        if (gap == 31) { if (has_x2) ld_x(s_p2, 1, 1); ld_y(s_0, 1, 1); }
#endif
        SCHED_FENCE();
      }
    }
    DKV3_STAMP(i, 1)
    // outside the steady loop the paths that join behind this iteration keep the score sets in different registers, and hipcc's
    // copies would read the chains' last results right behind the (opaque) MFMAs: rows 16 .. 31 of the last tile came out one
    // k-step short.  Routing the sets through this statement puts the copies behind 2 x 16 wait states.
    if constexpr (!STEADY) asm volatile("s_nop 15\n\ts_nop 15" : "+v"(AS), "+v"(AD));
    // requests are retired in order: tile i + 3 must have landed, the 5 + 5 of tiles i + 4 and i + 5 may stay in flight (towards
    // the end of the sweep fewer were made)
#ifndef LCV_DKV3_NO_DMA
    if (STEADY || i + 5 < nt) asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
    else if (i + 4 < nt) asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
    else lcv_dma_wait_all();
#endif
    DKV3_STAMP(i, 2)
    __builtin_amdgcn_s_barrier();
    DKV3_STAMP(i, 3)
    s_m1 = s_0; s_0 = s_p1; s_p1 = s_p2; s_p2 = adv(s_p2); s_p5 = adv(s_p5);
    DKV3_STAMP(i, 4)
  };

  PackedTile pa, pb;
  // ---- pipeline prologue: S / dP of tile 0 -> set A, its vector work in the open -> pa, S / dP of tile 1 -> set B ----
  ld_consts(0, sA, dA);
#pragma unroll
  for (int i = 0; i < PD; ++i) ld_x(0, i, i);
  phase_x(0, sA, dA);
  // (the only place where vector instructions read MFMA results right behind the MFMAs: asm is opaque to the hazard recogniser,
  // so the wait states are spelled out - 8 x 16 cycles, once per workgroup)
  asm volatile("s_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15" ::: "memory");
#pragma unroll
  for (int i = 0; i < 18; ++i) valu_step(i, pa, sA, dA);
  if (nt > 1) {
    ld_consts(STAGE, sB, dB);
#pragma unroll
    for (int i = 0; i < PD; ++i) ld_x(STAGE, i, i);
    phase_x(STAGE, sB, dB);
  }
  // what iteration 1 expects: constants and first k-steps of tile 2 (-> set A), first groups of tile 0
  if (nt > 2) {
    ld_consts(2 * STAGE, sA, dA);
#pragma unroll
    for (int i = 0; i < PD; ++i) ld_x(2 * STAGE, i, i);
  }
#pragma unroll
  for (int g = 0; g < PD; ++g) ld_y(0, g, g);
  {
    const int nfull = (int)(p.Nq / QT);                    // tiles 0 .. nfull - 1 are full
    int i = 1;
    for (; i + 6 < nfull; i += 2) {                       // steady state: tiles i + 5 and i + 6 exist and are full
      merged(i, pa, pb, sB, dB, sA, dA, std::true_type{});
      merged(i + 1, pb, pa, sA, dA, sB, dB, std::true_type{});
    }
    asm volatile("s_nop 15\n\ts_nop 15" : "+v"(sA), "+v"(dA), "+v"(sB), "+v"(dB));   // (the same at the steady loop's exit)
    for (; i + 1 < nt; i += 2) {                          // the last few tiles: the same iteration with its edge conditions
      merged(i, pa, pb, sB, dB, sA, dA, std::false_type{});
      merged(i + 1, pb, pa, sA, dA, sB, dB, std::false_type{});
    }
    if (i < nt) {
      merged(i, pa, pb, sB, dB, sA, dA, std::false_type{});
      phase_y(s_m1, pb);                                  // the last tile's products: nothing left to overlap
    } else {
      phase_y(s_m1, pa);
    }
  }

#ifdef LCV_DKV3_STAMPS
  if (dbg_on && lane_now() == 0)
    for (int i = 0; i < 64; ++i)
      g_dkv3_dbg[(wave ? 64 : 0) + i] = *reinterpret_cast<AS3X unsigned long long*>(lds + 7 * STAGE + wave * 1024 + i * 8);
#endif
  // ---- epilogue: acc[d][e] = dX^T[dim = 32 d + (e & 3) + 8 (e >> 2) + 4 h][key = lane & 31] ----
  const int ln = lane_now();
  const int r_l = ln & 31, h_l = ln >> 5;
  const int64_t krow = key0 + wave * 32 + r_l;
  if (krow < p.Nk) {
    bf16_t* dkp = p.dk + b * p.dk_sb + krow * p.dk_sn + (int64_t)head * p.dk_sh;
    bf16_t* dvp = p.dv + b * p.dv_sb + krow * p.dv_sn + (int64_t)head * p.dv_sh;
#pragma unroll
    for (int d = 0; d < 4; ++d)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int col = 32 * d + 8 * i + 4 * h_l;
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

// called by lcv_attn_bwd (attn_bwd.hip) when scale * log2(e) == 1 and LCV_ATTN_BWD_DKV=3
int attn_bwd_dkv3_launch(const void* q, const void* k, const void* v, const void* d_o, const float* consts,
                         void* dk, void* dv, int accumulate_kv, int64_t B, int64_t H, int64_t Nq, int64_t Nk, int64_t q_sb,
                         int64_t q_sn, int64_t q_sh, int64_t k_sb, int64_t k_sn, int64_t k_sh, int64_t v_sb, int64_t v_sn,
                         int64_t v_sh, int64_t o_sb, int64_t o_sn, int64_t o_sh, int64_t dk_sb, int64_t dk_sn, int64_t dk_sh,
                         int64_t dv_sb, int64_t dv_sn, int64_t dv_sh, float scale, hipStream_t s) {
  AttnBwdDkv3Params p;
  p.q = (const bf16_t*)q; p.k = (const bf16_t*)k; p.v = (const bf16_t*)v; p.d_o = (const bf16_t*)d_o;
  p.consts = consts; p.dk = (bf16_t*)dk; p.dv = (bf16_t*)dv; p.Nq = Nq; p.Nk = Nk; p.H = (int)H;
  p.q_sb = q_sb; p.q_sn = q_sn; p.q_sh = q_sh; p.k_sb = k_sb; p.k_sn = k_sn; p.k_sh = k_sh;
  p.v_sb = v_sb; p.v_sn = v_sn; p.v_sh = v_sh; p.o_sb = o_sb; p.o_sn = o_sn; p.o_sh = o_sh;
  p.dk_sb = dk_sb; p.dk_sn = dk_sn; p.dk_sh = dk_sh; p.dv_sb = dv_sb; p.dv_sn = dv_sn; p.dv_sh = dv_sh;
  p.scale = scale; p.accumulate_kv = accumulate_kv;
#ifdef LCV_DKV3_STAMPS
  const size_t lds = 7 * (2 * 32 * 256 + 2 * 32 * 4) + 4096;
#else
  const size_t lds = 7 * (2 * 32 * 256 + 2 * 32 * 4);
#endif
  static bool attr_set = false;
  if (!attr_set) {
    if (hipFuncSetAttribute((const void*)attn_bwd_dkv3_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) {
      lcv_set_error("attn_bwd: cannot raise dynamic LDS");
      return LCV_EDEVICE;
    }
    attr_set = true;
  }
  const dim3 grid((unsigned)((Nk + 127) / 128), (unsigned)H, (unsigned)B);
  hipLaunchKernelGGL(attn_bwd_dkv3_kernel, grid, dim3(256), lds, s, p);
  LCV_LAUNCH_CHECK("attn_bwd_dkv3");
  return LCV_OK;
}
