// Flash attention forward, 64 query rows per wave on ONE wave per SIMD (4 waves x 64 rows = the same 256-row block as
// attn_fwd_pipe_kernel, same 64-key tiles, same LDS images of K and V, same operand maps: S^T = K Q^T on
// v_mfma_f32_32x32x16_bf16 with the query on the lane, P^T straight from the accumulators as the B operand of O^T += V^T P^T).
//
// Why.  The ablation builds of the two-waves-per-SIMD kernel (scratch/attn_lab/build_pipe.sh ablation, profiles/r03_attn_fwd_lab.md)
// price its instruction classes at: K fragment reads 22 %, vector work 16 %, V fragment reads 14 %, LDS-DMA requests 10 % of the
// launch, against an MFMA-only loop that runs at 0.84 of the nominal peak.  Every wave of that kernel reads the WHOLE K and V tile
// from LDS for 32 query rows: 8 x 32 KiB = 256 KiB of LDS reads per 64 keys, ~2 000 of an iteration's ~3 000 cycles of LDS pipe.
// With 64 rows per wave a K fragment feeds two score MFMAs and a V^T fragment two PV MFMAs: half the LDS reads per MFMA, Q never
// leaves registers.  The price is the register file of a whole SIMD for one wave (O 128 + Q 64 accumulator registers, two score
// sets 128 + packs + fragments in arch VGPRs), i.e. nothing else issues while this wave waits.
//
// Structure (the software pipeline of attn_fwd_pipe.hip with two query blocks nb = 0, 1 per wave):
//   iteration t:   phase 1   S(t+1) = K(t+1) Q^T      32 MFMA   ||  P(t) = exp2(S(t)) elements 0-23 of both blocks, sums, packs
//                  barrier   K(t+2), V(t+1) in LDS for every wave
//                  phase 2   O += V(t)^T P(t)^T       32 MFMA   ||  elements 24-31, row max of S(t+1), 8 LDS-DMA requests of
//                                                                   K(t+3) / V(t+2), first fragments of K(t+2)
//                  post      rare: rescale O, l and S(t+1) of a block whose running max grew by more than 2^RESCALE_THR
// MFMAs are issued from asm with the register class pinned (score sets in arch VGPRs, O in AGPRs, Q fragments as AGPR B
// operands): left alone hipcc puts every MFMA result of a > 256-register kernel into AGPRs (round 3, scratch/tried/attn_bwd_dkv3_r3_one_wave_per_simd.hip.txt).  asm is opaque
// to the hazard recogniser: see GUARD below and profiles/r03_attn_bwd_lab.md for the four ways that went wrong before.
#include "lcv_common.h"
#include <type_traits>

typedef __attribute__((address_space(3))) unsigned char lds_u8w;
#define AS3W __attribute__((address_space(3)))
#define SCHED_FENCE() __builtin_amdgcn_sched_barrier(0)

struct AttnFwdW64Params {
  const bf16_t* q;
  const bf16_t* k;
  const bf16_t* v;
  bf16_t* o;
  float* lse;
  int64_t Nq, Nk;
  int H;
  int64_t q_sb, q_sn, q_sh, k_sb, k_sn, k_sh, v_sb, v_sn, v_sh, o_sb, o_sn, o_sh;
  float scale;
  int gx, xcd_remap;
};

#define W64_RESCALE_THR 6.0f


__device__ __forceinline__ float w64_half_max(float v) {
  float a = v, b = v;
  asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 0" : "+v"(a), "+v"(b));
  return fmaxf(a, b);
}
__device__ __forceinline__ float w64_half_sum(float v) {
  float a = v, b = v;
  asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 0" : "+v"(a), "+v"(b));
  return a + b;
}
// gap instructions: asm volatile keeps their program order (attn_fwd_pipe.hip)
__device__ __forceinline__ float w_exp2(float x) { float y; asm volatile("v_exp_f32 %0, %1" : "=v"(y) : "v"(x)); return y; }
__device__ __forceinline__ float w_add(float a, float b) { float y; asm volatile("v_add_f32 %0, %1, %2" : "=v"(y) : "v"(a), "v"(b)); return y; }
__device__ __forceinline__ float w_max3(float a, float b, float c) { float y; asm volatile("v_max3_f32 %0, %1, %2, %3" : "=v"(y) : "v"(a), "v"(b), "v"(c)); return y; }
typedef __attribute__((ext_vector_type(2))) float f32x2w;
__device__ __forceinline__ f32x2w w_pk_add(f32x2w a, f32x2w b) { f32x2w y; asm volatile("v_pk_add_f32 %0, %1, %2" : "=v"(y) : "v"(a), "v"(b)); return y; }
__device__ __forceinline__ unsigned w_pack(float lo, float hi) { unsigned y; asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(y) : "v"(lo), "v"(hi)); return y; }

// GUARD: wait states in front of the MFMA wherever hipcc may have placed a register copy of one of its operands right before the
// statement (everywhere outside the straight-line steady loop, and the first MFMAs of every phase)
template <bool GUARD>
__device__ __forceinline__ void mfma_s(f32x16& c, const bf16x8& a, const bf16x8& b) {   // score chain link: VGPR accumulator, B in AGPRs
  if constexpr (GUARD) asm volatile("s_nop 3\n\tv_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(c) : "v"(a), "a"(b));
  else asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(c) : "v"(a), "a"(b));
}
template <bool GUARD>
__device__ __forceinline__ void mfma_s_first(f32x16& d, const f32x16& c, const bf16x8& a, const bf16x8& b) {   // D != C: the resident -max tuple survives
  if constexpr (GUARD) asm volatile("s_nop 3\n\tv_mfma_f32_32x32x16_bf16 %0, %2, %3, %1" : "=&v"(d) : "v"(c), "v"(a), "a"(b));
  else asm volatile("v_mfma_f32_32x32x16_bf16 %0, %2, %3, %1" : "=&v"(d) : "v"(c), "v"(a), "a"(b));
}
template <bool GUARD>
__device__ __forceinline__ void mfma_o(f32x16& c, const bf16x8& a, const bf16x8& b) {   // O^T accumulator in AGPRs
  if constexpr (GUARD) asm volatile("s_nop 3\n\tv_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(c) : "v"(a), "v"(b));
  else asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(c) : "v"(a), "v"(b));
}

// element j (0..31) of the 64 scores a lane holds for one tile and one query block: j < 16 -> key block 0, else key block 1
#define SCW(S, nb, j) (S[(j) >> 4][nb][(j) & 15])

__global__ __launch_bounds__(256) void attn_fwd_w64_kernel(const AttnFwdW64Params p) {
  constexpr int TILE = 64 * 256;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  lds_u8w* lds = (lds_u8w*)smem;  // K buffers 0, 1 | V buffers 0, 1, 2
  constexpr int V_REGION = 2 * TILE;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 31, h = lane >> 5;
  int qb, head;
  int64_t b;
  if (p.xcd_remap) {   // head-per-XCD block order (speed only): see attn_fwd.hip
    const int id = blockIdx.x;
    const int xcd = id & 7, j = id >> 3;
    const int pair = (j / p.gx) * 8 + xcd;
    qb = j - (j / p.gx) * p.gx;
    head = pair % p.H;
    b = pair / p.H;
  } else {
    qb = blockIdx.x; head = blockIdx.y; b = blockIdx.z;
  }
  const int64_t q0 = (int64_t)qb * 256 + wave * 64;
  auto lane_now = []() -> int { return (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)); };
  const int nt = (int)((p.Nk + 63) / 64);
  const bool ragged = (p.Nk & 63) != 0;
  const char* kbase_u = lcv_uniform_ptr(p.k + b * p.k_sb + (int64_t)head * p.k_sh);
  const char* vbase_u = lcv_uniform_ptr(p.v + b * p.v_sb + (int64_t)head * p.v_sh);

  // ---- LDS-DMA roles: wave w fills rows 16 w .. 16 w + 15 of a tile with four 1-KiB requests (scalar tile base + a constant
  // per-lane 32-bit byte offset: row 16 w + 4 i + (lane >> 4), swizzled 16-byte column) ----
  auto dma_row_of = [&](int ln, int i) { return 16 * wave + 4 * i + (ln >> 4); };
  auto dma_colb_of = [&](int ln, int i) {
    const int row = dma_row_of(ln, i);
    return 16 * ((ln & 15) ^ (((row & 3) << 2) | ((row >> 2) & 3)));
  };
  unsigned koff[4], voff[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    koff[i] = (unsigned)(dma_row_of(lane, i) * p.k_sn * 2 + dma_colb_of(lane, i));
    voff[i] = (unsigned)(dma_row_of(lane, i) * p.v_sn * 2 + dma_colb_of(lane, i));
  }
  auto last_off = [&](int i, int64_t sn) {   // the last tile's rows past Nk re-read the last key (their scores are masked)
    const int ln = lane_now();
    int64_t row = (int64_t)(nt - 1) * 64 + dma_row_of(ln, i);
    if (row > p.Nk - 1) row = p.Nk - 1;
    return (unsigned)(row * sn * 2 + dma_colb_of(ln, i));
  };
  const unsigned lds_wave = (unsigned)(uintptr_t)lds + (unsigned)wave * 4096u;   // this wave's 4 KiB slice of every tile
  // piece i (0..3) of tile `tile` of K (which = 0) or V (which = 1) to the buffer at LDS byte offset `dst_tile`
  // (which / i are plain ints that fold once the calling loop is unrolled: ONE copy of this body per gap, or the unroller gives up)
  auto dma_one = [&](int which, int i, int dst_tile, int tile, bool known_full) __attribute__((always_inline)) {
    const int64_t sn = which ? p.v_sn : p.k_sn;
    const char* base = which ? vbase_u : kbase_u;
    unsigned off = which ? voff[i] : koff[i];
    if (!known_full && tile == nt - 1) off = last_off(i, sn);
    else base += (int64_t)tile * (128 * sn);
    // (M0 carries the LDS destination; nothing else in this kernel uses it, so it is declared clobbered instead of saved and
    // restored around every request: two scalar instructions less per request on a wave whose issue slots are the budget)
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1"
                 :: "v"(off), "s"(base), "s"(lds_wave + (unsigned)dst_tile + 1024u * i) : "memory", "m0");
  };
  auto dma_tile = [&](int which, int dst_tile, int tile) __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < 4; ++i) dma_one(which, i, dst_tile, tile, false);
  };
  constexpr int KOP = 0, VOP = 1;
  auto dma_wait_and_barrier = [&]() { asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory"); };

  // ---- per-lane LDS read offsets (the images of attn_fwd_pipe.hip) ----
  int k_off[8];
  int v_off[2][4];
  auto set_read_offsets = [&](int ln, int slot) {
    const int rr = ln & 31, hh = ln >> 5;
    const int kfz = ((rr & 3) << 2) | ((rr >> 2) & 3);
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) k_off[ks] = 256 * rr + 16 * ((2 * ks + hh) ^ kfz);
    const int q4 = (ln >> 2) & 3, p4 = ln & 3, g1 = (ln >> 4) & 1;
#pragma unroll
    for (int half = 0; half < 2; ++half)
#pragma unroll
      for (int d = 0; d < 4; ++d)
        v_off[half][d] = (int)(unsigned)(uintptr_t)lds + V_REGION + slot * TILE + 256 * (4 * hh + 8 * half + q4) + 8 * (p4 & 1) + 64 * (d ^ q4) +
                         16 * ((2 * g1 + (p4 >> 1)) ^ (hh + 2 * half));   // ABSOLUTE LDS byte address: no base add per read
  };
  set_read_offsets(lane, 2);   // (V slot 2: iteration 0 rotates the offsets to slot 0)
  auto read_k = [&](const lds_u8w* kb, int f) __attribute__((always_inline)) -> bf16x8 {   // K fragment f: k-step f >> 1, key block f & 1
    return *reinterpret_cast<const AS3W bf16x8*>(kb + (f & 1) * 32 * 256 + k_off[f >> 1]);
  };
  auto read_v = [&](int g) __attribute__((always_inline)) -> bf16x8 {   // V^T fragment g: k-step g >> 2, dim block g & 3, of the slot the offsets point at
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((AS3W s16x4*)(uintptr_t)(unsigned)(v_off[0][g & 3] + 4096 * (g >> 2)));
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((AS3W s16x4*)(uintptr_t)(unsigned)(v_off[1][g & 3] + 4096 * (g >> 2)));
    return __builtin_bit_cast(bf16x8, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
  };

  // Q fragments (B operands of the score MFMAs), resident in AGPRs for the whole sweep: qf[nb][ks] = Q[q0 + 32 nb + r][16 ks + 8 h ..]
  bf16x8 qf[2][8];
#pragma unroll
  for (int nb = 0; nb < 2; ++nb) {
    int64_t qrow = q0 + 32 * nb + r;
    if (qrow > p.Nq - 1) qrow = p.Nq - 1;
    const bf16_t* qp = p.q + b * p.q_sb + qrow * p.q_sn + (int64_t)head * p.q_sh + 8 * h;
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) qf[nb][ks] = *reinterpret_cast<const bf16x8*>(qp + 16 * ks);
  }
#pragma unroll
  for (int nb = 0; nb < 2; ++nb)
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) asm volatile("" : "+a"(qf[nb][ks]));   // into AGPRs here, far from the first MFMA that reads them

  f32x16 oacc[2][4];   // [query block][dim block]
#pragma unroll
  for (int nb = 0; nb < 2; ++nb)
#pragma unroll
    for (int d = 0; d < 4; ++d) {
#pragma unroll
      for (int e = 0; e < 16; ++e) oacc[nb][d][e] = 0.f;
      asm volatile("" : "+a"(oacc[nb][d]));
    }
  float m_run[2] = {0.f, 0.f}, l_run[2] = {0.f, 0.f};
  f32x16 minit[2];   // -m_run of a query block in every element: the C operand of its chains' first MFMAs
#pragma unroll
  for (int nb = 0; nb < 2; ++nb)
#pragma unroll
    for (int e = 0; e < 16; ++e) minit[nb][e] = 0.f;
  f32x16 sa[2][2], sb[2][2];   // score sets A and B: [key block][query block]

  auto mask_last = [&](f32x16 (&s)[2][2]) __attribute__((always_inline)) {   // scores of the ragged last tile past Nk -> -inf
    const int valid = (int)(p.Nk - (int64_t)(nt - 1) * 64);
    const int hh_ = lane_now() >> 5;
#pragma unroll
    for (int nb = 0; nb < 2; ++nb)
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int key = (e & 3) + 8 * (e >> 2) + 4 * hh_;
        if (key >= valid) s[0][nb][e] = -INFINITY;
        if (key + 32 >= valid) s[1][nb][e] = -INFINITY;
      }
  };
  // all accumulators through one statement that carries wait states: whatever hipcc does with them next (copies for a join,
  // v_accvgpr_read for the rescale) sits behind the MFMAs' write-back
  auto fence_o = [&]() __attribute__((always_inline)) {
    asm volatile("s_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15"
                 : "+a"(oacc[0][0]), "+a"(oacc[0][1]), "+a"(oacc[0][2]), "+a"(oacc[0][3]), "+a"(oacc[1][0]), "+a"(oacc[1][1]),
                   "+a"(oacc[1][2]), "+a"(oacc[1][3]));
  };
  auto fence_s = [&](f32x16 (&s)[2][2]) __attribute__((always_inline)) {
    asm volatile("s_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15" : "+v"(s[0][0]), "+v"(s[0][1]), "+v"(s[1][0]), "+v"(s[1][1]));
  };
  // row max of a query block's score tile relative to its running max, and the (rare) rescale it may trigger
  auto settle = [&](f32x16 (&s)[2][2], auto nb_c, float mx, bool first, bool need) __attribute__((always_inline)) {
    constexpr int nb = decltype(nb_c)::value;
    if (need || first) {   // need: some lane's row max exceeds the threshold (a ballot the caller took inside an MFMA gap)
      fence_o();
      const float d = first ? mx : fmaxf(mx, 0.f);
      const float alpha = __builtin_amdgcn_exp2f(-d);
      m_run[nb] += d;
      l_run[nb] *= alpha;
#pragma unroll
      for (int dd = 0; dd < 4; ++dd)
#pragma unroll
        for (int e = 0; e < 16; ++e) oacc[nb][dd][e] *= alpha;
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        s[0][nb][e] -= d;
        s[1][nb][e] -= d;
        // an in-place update as far as hipcc can tell: a fresh splat here made it copy the whole tuple (8 v_mov_b64 per query
        // block) on the COMMON path of every iteration to merge the two versions
        asm volatile("v_mov_b32 %0, %1" : "+v"(minit[nb][e]) : "v"(-m_run[nb]));
      }
      fence_o();
    }
  };

#ifndef W64_PD
#define W64_PD 2
#endif
#ifndef W64_DMA_STRIDE
#define W64_DMA_STRIDE 1
#endif
  constexpr int PD = W64_PD, RING = PD + 1;   // fragments are requested PD fragments (= 2 PD MFMAs) ahead of their first use
  bf16x8 kfr[RING], vfr[RING];
  int v_slot = 2;
  auto next_v_slot = [&]() __attribute__((always_inline)) -> int {
    v_slot = (v_slot == 2) ? 0 : v_slot + 1;
    return (v_slot == 0) ? -2 * TILE : TILE;
  };

  // ---- prologue: K(0), V(0), K(1), V(1) requested; S(0) computed plainly and settled; then K(2) into K(0)'s buffer ----
  dma_tile(KOP, 0, 0);                  // (the launcher guarantees nt >= 6)
  dma_tile(VOP, V_REGION, 0);
  dma_tile(KOP, TILE, 1);
  dma_tile(VOP, V_REGION + TILE, 1);
  dma_wait_and_barrier();
  {
#pragma unroll
    for (int kb_ = 0; kb_ < 2; ++kb_)
#pragma unroll
      for (int nb = 0; nb < 2; ++nb)
#pragma unroll
        for (int e = 0; e < 16; ++e) sa[kb_][nb][e] = 0.f;
#pragma unroll
    for (int f = 0; f < 16; ++f) {
      const bf16x8 a = read_k(lds, f);
      mfma_s<true>(sa[f & 1][0], a, qf[0][f >> 1]);
      mfma_s<true>(sa[f & 1][1], a, qf[1][f >> 1]);
    }
    fence_s(sa);
    auto first_settle = [&](auto nb_c) __attribute__((always_inline)) {
      constexpr int nb = decltype(nb_c)::value;
      float mx = sa[0][nb][0];
#pragma unroll
      for (int e = 1; e < 16; ++e) mx = fmaxf(mx, sa[0][nb][e]);
#pragma unroll
      for (int e = 0; e < 16; ++e) mx = fmaxf(mx, sa[1][nb][e]);
      settle(sa, nb_c, w64_half_max(mx), true, true);
    };
    first_settle(std::integral_constant<int, 0>{});
    first_settle(std::integral_constant<int, 1>{});
  }
  __syncthreads();                        // every wave has read K(0)
  dma_tile(KOP, 0, 2);                  // K(2) -> K buffer 0; waited for at the barrier of iteration 0
#pragma unroll
  for (int i = 0; i < PD; ++i) kfr[i] = read_k(lds + TILE, i);   // first fragments of K(1): what an iteration expects

  // ---- one pipelined iteration.  PAR = t & 1: K(t+1) in K buffer PAR ^ 1, K(t+2) in buffer PAR, K(t+3) requested into buffer
  // PAR ^ 1 after the barrier; V(t) in slot t % 3, V(t+2) requested into slot (t + 2) % 3.  c = S(t), settled; n receives S(t+1).
  // STEADY: tiles up to t + 3 exist and are full, no edge condition is evaluated and the code is straight-line.
  auto iteration = [&](const int t, auto par_c, auto steady_c, f32x16 (&c)[2][2], f32x16 (&n)[2][2]) __attribute__((always_inline)) {
    constexpr int PAR = decltype(par_c)::value;
    constexpr bool STEADY = decltype(steady_c)::value;
    const lds_u8w* kb = lds + (PAR ^ 1) * TILE;      // K(t+1)
    const lds_u8w* kb_next = lds + PAR * TILE;        // K(t+2)
    const bool has_k3 = STEADY || t + 3 < nt;
    const bool has_v2 = STEADY || t + 2 < nt;
    const int v_delta = next_v_slot();                // v_slot == t % 3 from here on
    const int v_dst = V_REGION + ((v_slot == 0) ? 2 : v_slot - 1) * TILE;   // slot (t + 2) % 3
    f32x2w psum[2];       // row sums of P(t), even / odd elements apart (v_pk_add_f32: one instruction per pair)
    f32x2w ex[2][16];     // P(t) in fp32: pair m = elements (2m, 2m + 1)
    unsigned pw[2][16];   // P(t) as packed bf16 pairs: word m = elements (2m, 2m + 1)
    SCHED_FENCE();
    // ---------------- phase 1: 32 score MFMAs of tile t+1; exp2 / sums / packs of elements 0..23 of both query blocks ----------
#pragma unroll
    for (int i = 0; i < 32; ++i) {
      const int f = i >> 1, nb = i & 1;               // fragment f = (k-step f >> 1, key block f & 1) feeds MFMAs 2 f, 2 f + 1
      if (nb == 0) {
        if (f + PD < 16) kfr[(f + PD) % RING] = read_k(kb, f + PD);
        if (f >= 16 - PD) vfr[f - (16 - PD)] = read_v(f - (16 - PD));   // first fragments of V(t) (landed since the last barrier)
      }
      if (f < 2) {
        if (STEADY && i >= 1) mfma_s_first<false>(n[f & 1][nb], minit[nb], kfr[f % RING], qf[nb][0]);
        else mfma_s_first<true>(n[f & 1][nb], minit[nb], kfr[f % RING], qf[nb][0]);
      } else {
        if (STEADY) mfma_s<false>(n[f & 1][nb], kfr[f % RING], qf[nb][f >> 1]);
        else mfma_s<true>(n[f & 1][nb], kfr[f % RING], qf[nb][f >> 1]);
      }
      SCHED_FENCE();
      // gaps 10..17: one of the eight V read offsets moves to tile t's slot (no V read is in flight between gap 0 and gap 27)
      if (i >= 10 && i < 18) asm volatile("v_add_u32 %0, %1, %0" : "+v"(v_off[(i - 10) >> 2][(i - 10) & 3]) : "s"(v_delta));
      // exps of this gap: indices [e_lo, e_hi) of 48 (index -> query block idx & 1, element idx >> 1); sums / packs trail one gap
      const int e_lo = (3 * i + 1) / 2, e_hi = (3 * (i + 1) + 1) / 2;
      const int a_lo = i ? (3 * (i - 1) + 1) / 2 : 0, a_hi = i ? e_lo : 0;
#pragma unroll
      for (int u = 0; u < 2; ++u) {                   // (at most two per gap; fixed trip count so that the loop unrolls)
        const int x = e_lo + u;
        if (x < e_hi) ex[x & 1][x >> 2][(x >> 1) & 1] = w_exp2(SCW(c, x & 1, x >> 1));
      }
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const int x = a_lo + u;
        if (x < a_hi) {
          const int b_ = x & 1, j = x >> 1;
          if (j & 1) {                                  // pair j >> 1 of block b_ is complete
            pw[b_][j >> 1] = w_pack(ex[b_][j >> 1][0], ex[b_][j >> 1][1]);
            if (j == 3) psum[b_] = w_pk_add(ex[b_][0], ex[b_][1]);
            else if (j > 3) psum[b_] = w_pk_add(psum[b_], ex[b_][j >> 1]);
          }
        }
      }
      SCHED_FENCE();
    }
    if constexpr (!STEADY) {
      fence_s(n);
      if (t + 1 == nt - 1 && ragged) mask_last(n);   // scalar branch, taken once
    }
    // the one barrier: K(t+2) and V(t+1) are in LDS for every wave; every wave has finished reading K(t+1) and V(t-1)
    dma_wait_and_barrier();
    // ---------------- phase 2: 32 PV MFMAs of tile t; the rest of P(t); row max of S(t+1); next requests and fragments --------
    float mxa[2] = {0.f, 0.f}, mxb[2] = {0.f, 0.f}, mxh[2] = {0.f, 0.f};
    bool need[2] = {false, false};
    SCHED_FENCE();
#pragma unroll
    for (int j = 0; j < 32; ++j) {
      const int g = j >> 1, nb = j & 1;               // fragment g = (k-step g >> 2, dim block g & 3) feeds MFMAs 2 g, 2 g + 1
      if (nb == 0) {
        if (g + PD < 16) vfr[(g + PD) % RING] = read_v(g + PD);
        if (g >= 16 - PD) kfr[g - (16 - PD)] = read_k(kb_next, g - (16 - PD));   // first fragments of K(t+2)
      }
      const int kk = g >> 2;
      const u32x4 pbw = {pw[nb][4 * kk], pw[nb][4 * kk + 1], pw[nb][4 * kk + 2], pw[nb][4 * kk + 3]};
      if (STEADY && j >= 2) mfma_o<false>(oacc[nb][g & 3], vfr[g % RING], __builtin_bit_cast(bf16x8, pbw));
      else mfma_o<true>(oacc[nb][g & 3], vfr[g % RING], __builtin_bit_cast(bf16x8, pbw));
      SCHED_FENCE();
      // the eight LDS-DMA requests of this iteration: K(t+3) into K(t+1)'s buffer, V(t+2) into V(t-1)'s slot (both free since the
      // barrier above); waited for at the next barrier, a whole iteration away
      if (j % W64_DMA_STRIDE == 0 && j / W64_DMA_STRIDE < 8) {
        const int dj = j / W64_DMA_STRIDE;
        if (dj < 4) { if (has_k3) dma_one(KOP, dj & 3, (PAR ^ 1) * TILE, t + 3, STEADY); }
        else { if (has_v2) dma_one(VOP, dj & 3, v_dst, t + 2, STEADY); }
      }
      // row max of S(t+1) first (gaps 0..15: per query block 16 max3 over its 32 values, two chains of 8), then the exchange with
      // the partner half and the ballot in gaps 16..19: nothing but two scalar branches is left behind the phase
      if (j < 16) {
        const int b_ = j & 1, s_ = j >> 1;            // step 0..7 of block b_
        if (s_ == 0) {
          mxa[b_] = w_max3(n[0][b_][0], n[0][b_][1], n[0][b_][2]);
          mxb[b_] = w_max3(n[1][b_][0], n[1][b_][1], n[1][b_][2]);
        } else if (s_ < 7) {
          mxa[b_] = w_max3(mxa[b_], n[0][b_][2 * s_ + 1], n[0][b_][2 * s_ + 2]);
          mxb[b_] = w_max3(mxb[b_], n[1][b_][2 * s_ + 1], n[1][b_][2 * s_ + 2]);
        } else {
          mxa[b_] = w_max3(mxa[b_], n[0][b_][15], mxb[b_]);
          mxb[b_] = w_max3(mxa[b_], n[1][b_][15], n[1][b_][14]);   // (the full maximum: chain a folded in)
        }
      }
      if (j == 16 || j == 17) mxh[j & 1] = w64_half_max(mxb[j & 1]);
      if (j == 18 || j == 19) need[j & 1] = __builtin_amdgcn_ballot_w64(mxh[j & 1] > W64_RESCALE_THR) != 0ull;
      // quarter 3 of P(t): index 47 (query block 1, element 23) was exp'ed in the last gap of phase 1; elements 24..31 one exp per
      // gap in gaps 6..21, the sum / pack two gaps later (the same block's next turn); all packs exist before the k-step 3 MFMAs
      if (j == 0) {
        psum[1] = w_pk_add(psum[1], ex[1][11]);
        pw[1][11] = w_pack(ex[1][11][0], ex[1][11][1]);
      }
      if (j >= 6 && j < 22) {
        const int b_ = j & 1, el = 24 + ((j - 6) >> 1);
        ex[b_][el >> 1][el & 1] = w_exp2(SCW(c, b_, el));
      }
      if (j >= 8 && j < 24) {
        const int b_ = j & 1, el = 24 + ((j - 8) >> 1);
        if (el & 1) {
          psum[b_] = w_pk_add(psum[b_], ex[b_][el >> 1]);
          pw[b_][el >> 1] = w_pack(ex[b_][el >> 1][0], ex[b_][el >> 1][1]);
        }
      }
      if (j == 24 || j == 25) psum[j & 1][0] = w_add(psum[j & 1][0], psum[j & 1][1]);
      if (j == 26 || j == 27) l_run[j & 1] = w_add(l_run[j & 1], psum[j & 1][0]);
      SCHED_FENCE();
    }
    settle(n, std::integral_constant<int, 0>{}, mxh[0], false, need[0]);
    settle(n, std::integral_constant<int, 1>{}, mxh[1], false, need[1]);
  };

  // last tile: nothing left to overlap with; c = S(nt - 1), settled; its V tile landed before the last barrier
  auto final_tile = [&](f32x16 (&c)[2][2]) __attribute__((always_inline)) {
    const int v_delta = next_v_slot();
#pragma unroll
    for (int i = 0; i < 8; ++i) v_off[i >> 2][i & 3] += v_delta;
    unsigned pw[2][16];
#pragma unroll
    for (int nb = 0; nb < 2; ++nb) {
      float psum = 0.f;
      float ex[32];
#pragma unroll
      for (int j = 0; j < 32; ++j) {
        ex[j] = __builtin_amdgcn_exp2f(SCW(c, nb, j));
        psum += ex[j];
      }
      l_run[nb] += psum;
#pragma unroll
      for (int m = 0; m < 16; ++m) pw[nb][m] = w_pack(ex[2 * m], ex[2 * m + 1]);
    }
#pragma unroll
    for (int g = 0; g < 16; ++g) {
      const bf16x8 vf_ = read_v(g);
      const int kk = g >> 2;
#pragma unroll
      for (int nb = 0; nb < 2; ++nb) {
        const u32x4 pbw = {pw[nb][4 * kk], pw[nb][4 * kk + 1], pw[nb][4 * kk + 2], pw[nb][4 * kk + 3]};
        mfma_o<true>(oacc[nb][g & 3], vf_, __builtin_bit_cast(bf16x8, pbw));
      }
    }
    fence_o();
  };

  {
    using P0 = std::integral_constant<int, 0>;
    using P1 = std::integral_constant<int, 1>;
    int t = 0;
    for (; t + 5 < nt; t += 2) {           // steady state: tiles up to t + 4 exist and are FULL (t + 4 is not the last, maybe ragged, one)
      iteration(t, P0{}, std::true_type{}, sa, sb);
      iteration(t + 1, P1{}, std::true_type{}, sb, sa);
    }
    fence_o(); fence_s(sa); fence_s(sb);
    for (; t + 1 <= nt - 2; t += 2) {
      iteration(t, P0{}, std::false_type{}, sa, sb);
      iteration(t + 1, P1{}, std::false_type{}, sb, sa);
    }
    if (t == nt - 2) iteration(t, P0{}, std::false_type{}, sa, sb);
    fence_o(); fence_s(sa); fence_s(sb);
    set_read_offsets(lane_now(), v_slot);   // (fresh copies for the last tile)
    if ((nt - 1) & 1) final_tile(sb);
    else final_tile(sa);
  }

  // ---- epilogue ----
  const int lane_l = lane_now();
  const int r_l = lane_l & 31, h_l = lane_l >> 5;
#pragma unroll
  for (int nb = 0; nb < 2; ++nb) {
    const float l_tot = w64_half_sum(l_run[nb]);
    const float inv = 1.0f / l_tot;
    const int64_t qrow = q0 + 32 * nb + r_l;
    if (qrow < p.Nq) {
      bf16_t* op = p.o + b * p.o_sb + qrow * p.o_sn + (int64_t)head * p.o_sh;
#pragma unroll
      for (int d = 0; d < 4; ++d)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          u16x4 pk;
#pragma unroll
          for (int e = 0; e < 4; ++e) pk[e] = f2bf(oacc[nb][d][4 * i + e] * inv);
          *reinterpret_cast<u16x4*>(op + 32 * d + 8 * i + 4 * h_l) = pk;
        }
      if (p.lse && h_l == 0) p.lse[(b * p.H + head) * p.Nq + qrow] = m_run[nb] * p.scale + __logf(l_tot);
    }
  }
}

// called by lcv_attn_fwd (attn_fwd.hip) for unit-scale self-attention with Nk > 512: the default since round 3 (LCV_ATTN_FWD_W64=0
// selects attn_fwd_pipe.hip instead)
int attn_fwd_w64_launch(const void* q, const void* k, const void* v, void* o, float* lse, int64_t B, int64_t H, int64_t Nq,
                        int64_t Nk, int64_t q_sb, int64_t q_sn, int64_t q_sh, int64_t k_sb, int64_t k_sn, int64_t k_sh,
                        int64_t v_sb, int64_t v_sn, int64_t v_sh, int64_t o_sb, int64_t o_sn, int64_t o_sh, float scale,
                        int xcd_ok, hipStream_t s) {
  AttnFwdW64Params p;
  p.q = (const bf16_t*)q; p.k = (const bf16_t*)k; p.v = (const bf16_t*)v; p.o = (bf16_t*)o; p.lse = lse;
  p.Nq = Nq; p.Nk = Nk; p.H = (int)H;
  p.q_sb = q_sb; p.q_sn = q_sn; p.q_sh = q_sh; p.k_sb = k_sb; p.k_sn = k_sn; p.k_sh = k_sh;
  p.v_sb = v_sb; p.v_sn = v_sn; p.v_sh = v_sh; p.o_sb = o_sb; p.o_sn = o_sn; p.o_sh = o_sh;
  p.scale = scale;
  const unsigned gx = (unsigned)((Nq + 255) / 256);
  p.gx = (int)gx;
  p.xcd_remap = (xcd_ok && (B * H) % 8 == 0 && gx >= 8) ? 1 : 0;
  const size_t lds = 5 * 64 * 256;   // K x2, V x3
  // (function-local static: initialised once, thread-safe)
  static const bool attr_ok = !(hipFuncSetAttribute((const void*)attn_fwd_w64_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess);
  if (!attr_ok) {
      lcv_set_error("attn_fwd_w64: cannot raise dynamic LDS");
      return LCV_EDEVICE;
  }
  const dim3 grid = p.xcd_remap ? dim3(gx * (unsigned)(H * B)) : dim3(gx, (unsigned)H, (unsigned)B);
  hipLaunchKernelGGL(attn_fwd_w64_kernel, grid, dim3(256), lds, s, p);
  LCV_LAUNCH_CHECK("attn_fwd_w64");
  return LCV_OK;
}
