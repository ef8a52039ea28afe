// Flash attention forward, software-pipelined form, for the self-attention path of the DiT (q pre-scaled into log2 units:
// scale * log2(e) == 1, head_dim 128, Nk >= 256).  Same geometry, operand maps, LDS image and block order as
// attn_fwd_kernel<8, 0, false, 3> (attn_fwd.hip): 8 waves x 32 query rows, 64-key tiles, S^T = K Q^T on
// v_mfma_f32_32x32x16_bf16 with the query on the lane, P^T straight from the accumulator as the B operand of O^T += V^T P^T,
// K / V tiles by LDS-DMA.  What changes is the ORDER of each wave's instruction stream.
//
// Why.  attn_fwd_kernel runs per tile  [16 MFMA QK^T] -> [~230 vector instructions of softmax] -> [16 MFMA PV]  and leaves the
// overlap of one wave's softmax with the other wave's MFMAs to the hardware.  Measured on this chip (scratch/coexec/coexec3.hip,
// two waves per SIMD, the vector work of one tile beside 32 MFMAs): phases in lockstep 1 208 ns per iteration, staggered by
// half a period 1 212, s_setprio around the vector block 1 132-1 145 — but the SAME instructions spread between the MFMAs of
// one stream: 941-962 ns, against a matrix-pipe floor of ~890.  An MFMA holds the SIMD's vector issue for 8 of its 32 cycles;
// the other 24 take one transcendental (8) plus up to four plain instructions (4 each) — IF they come from the wave that owns
// the MFMA.  So every wave carries its own softmax in its own MFMA gaps:
//
//   iteration t:   phase 1   S(t+1) = K(t+1) Q^T      16 MFMA   ||  P(t) = exp2(S(t)) quarters 0-2, row sums, bf16 packing
//                  barrier   (the only one: K(t+2) and V(t+1), requested half an iteration ago, are in LDS for every wave)
//                  phase 2   O += V(t)^T P(t)^T       16 MFMA   ||  quarter 3 of P(t), row max of S(t+1), cross-half max, the
//                                                                   LDS-DMA of K(t+3) / V(t+2), the first fragments of K(t+2)
//                  post      rare: rescale O, l and S(t+1) when the running max grew by more than 2^RESCALE_THR
//
// The score accumulators of tile t+1 start at -running_max (a resident 16-register tuple read as the C operand of the first MFMA
// of each chain), so P = exp2(accumulator) with no subtraction; two named score register sets (A / B) swap roles every
// iteration (loop unrolled by two).  LDS (144 KiB): two K buffers, THREE V buffers (the V rotation is a scalar added to the eight
// transposed-read offsets once per iteration), and every wave's 32 Q rows fragment-major (8 KiB per wave: one base register +
// 1024 ks; the 32 registers of Q fragments would not fit beside two score sets, O and the resident tuple).
// The barrier sits BETWEEN the phases, where nothing depends on data that is still in flight: the stamped build of the first
// version (barrier at the end of the iteration, scratch/attn_lab/pipe_main.cpp) showed ~500 of ~2 950 cycles per iteration in
// [row max + exchange + vmcnt + barrier + first LDS reads of the next iteration] with the matrix pipe idle; now the fragments of
// the next phase are requested before the barrier / before the iteration ends and the reductions sit in MFMA gaps.
// Each gap's instructions are pinned (asm volatile + sched_barrier(0)): the order in this file IS the order in the binary.
#include "lcv_common.h"
#include <type_traits>

typedef __attribute__((address_space(3))) unsigned char lds_u8p;
typedef __attribute__((address_space(1))) void gbl_void_p;
typedef __attribute__((address_space(3))) void lds_void_p;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
#define AS3P __attribute__((address_space(3)))
#define SCHED_FENCE() __builtin_amdgcn_sched_barrier(0)

struct AttnFwdPipeParams {
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
  int prio_mode;   // how the two waves of a SIMD take turns in the issue arbitration (0 = none; the lab variants measured no gain)
};

#define PIPE_RESCALE_THR 6.0f

__device__ __forceinline__ float pipe_half_max(float v) {
  float a = v, b = v;
  asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 0" : "+v"(a), "+v"(b));
  return fmaxf(a, b);
}
__device__ __forceinline__ float pipe_half_sum(float v) {
  float a = v, b = v;
  asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 0" : "+v"(a), "+v"(b));
  return a + b;
}
__device__ __forceinline__ unsigned pack_bf16x2(float lo, float hi) {
  bf16x2_t t;
  t[0] = (__bf16)lo;
  t[1] = (__bf16)hi;
  return __builtin_bit_cast(unsigned, t);
}

// The vector instructions of the gaps are asm volatile ON PURPOSE: hipcc's instruction selection is free to hoist a pure
// builtin (it gathered all 24 exponentials of phase 1 behind the second MFMA), while volatile statements keep their program order
// among themselves and against sched_barrier(0).  Every result is consumed at least one gap later, so no statement needs a wait
// state inside it (a transcendental's result is not read by the next instruction, an MFMA operand not written just before it).
__device__ __forceinline__ float g_exp2(float x) { float y; asm volatile("v_exp_f32 %0, %1" : "=v"(y) : "v"(x)); return y; }
__device__ __forceinline__ float g_add(float a, float b) { float y; asm volatile("v_add_f32 %0, %1, %2" : "=v"(y) : "v"(a), "v"(b)); return y; }
__device__ __forceinline__ float g_max3(float a, float b, float c) { float y; asm volatile("v_max3_f32 %0, %1, %2, %3" : "=v"(y) : "v"(a), "v"(b), "v"(c)); return y; }
__device__ __forceinline__ unsigned g_pack(float lo, float hi) { unsigned y; asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(y) : "v"(lo), "v"(hi)); return y; }

// element j (0..31) of the 64 scores a lane holds for one tile: j < 16 -> s0[j], else s1[j - 16]; quarter q = j >> 3 is the
// B operand of PV k-step q
#define SC(S0, S1, j) ((j) < 16 ? S0[(j) & 15] : S1[(j) & 15])   /* rvalue */

__global__ __launch_bounds__(512) void attn_fwd_pipe_kernel(const AttnFwdPipeParams p) {
  constexpr int TILE = 64 * 256;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  lds_u8p* lds = (lds_u8p*)smem;  // K buffers 0, 1 | V buffers 0, 1, 2
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
  const int64_t q0 = (int64_t)qb * 256 + wave * 32;
  // Issue arbitration between the two waves of a SIMD is "priority, then age": left alone, waves 0-3 (older) win every phase,
  // reach the barrier early and idle there while waves 4-7 finish.  prio_mode 1 / 2: the halves swap priority every phase, so
  // between two barriers each half is the favoured one once; 3: waves 4-7 favoured throughout.
  const bool hi_half = wave >= 4;
  auto phase_prio = [&](int phase) {
    if (p.prio_mode == 1 || p.prio_mode == 2) {
      const bool fav = (phase == (p.prio_mode == 1 ? 1 : 2)) ? hi_half : !hi_half;
      if (fav) __builtin_amdgcn_s_setprio(1); else __builtin_amdgcn_s_setprio(0);
    }
  };
  if (p.prio_mode == 3 && hi_half) __builtin_amdgcn_s_setprio(1);
  const int nt_ = (int)((p.Nk + 63) / 64);
  const bf16_t* kbase = p.k + b * p.k_sb + (int64_t)head * p.k_sh;
  const bf16_t* vbase = p.v + b * p.v_sb + (int64_t)head * p.v_sh;

  // ---- LDS-DMA roles: wave w fills rows 8 w .. 8 w + 7 of a tile with two 1-KiB instructions ----
  // Source address = scalar base of the tile (SGPR pair, advanced one tile per issue by scalar adds) + a per-lane 32-bit byte
  // offset that never changes (row 8 w + 4 i + (lane >> 4), swizzled 16-byte column): no vector address arithmetic in the loop.
  // (named scalars, not arrays: a run-time choice between two arrays would send both to scratch)
  unsigned koff0, koff1, voff0, voff1;        // every full tile
  // The lane id is re-derived from the hardware (v_mbcnt) wherever a rare branch, the last tile or the epilogue needs it:
  // no register holds it across the loop, and what is computed from it cannot be hoisted in front of the loop (where output
  // pointers and edge-case offsets once sat in ~50 registers for the whole sweep and pushed loop values into scratch)
  auto lane_now = []() -> int { return (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)); };
  auto dma_row_of = [&](int ln, int i) { return 8 * wave + 4 * i + (ln >> 4); };
  auto dma_colb_of = [&](int ln, int i) {
    const int row = dma_row_of(ln, i);
    return 16 * ((ln & 15) ^ (((row & 3) << 2) | ((row >> 2) & 3)));
  };
  auto full_off = [&](int i, int64_t sn) { return (unsigned)(dma_row_of(lane, i) * sn * 2 + dma_colb_of(lane, i)); };
  auto last_off = [&](int i, int64_t sn) {
    const int ln = lane_now();
    int64_t row = (int64_t)(nt_ - 1) * 64 + dma_row_of(ln, i);
    if (row > p.Nk - 1) row = p.Nk - 1;
    return (unsigned)(row * sn * 2 + dma_colb_of(ln, i));
  };
  koff0 = full_off(0, p.k_sn); koff1 = full_off(1, p.k_sn); voff0 = full_off(0, p.v_sn); voff1 = full_off(1, p.v_sn);
  // scalar (SGPR) bases of this (batch, head)'s K and V rows: readfirstlane makes the uniformity provable, so the asm below
  // gets its "s" operands (a pointer hipcc cannot prove uniform would be handed over in VGPRs)
  auto uniform_ptr = [](const bf16_t* ptr) -> const char* {
    const unsigned long long v = (unsigned long long)ptr;
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
    return (const char*)(((unsigned long long)hi << 32) | lo);
  };
  const char* kbase_u = uniform_ptr(kbase);
  const char* vbase_u = uniform_ptr(vbase);
  const unsigned lds_wave = (unsigned)(uintptr_t)lds + (unsigned)wave * 2048u;   // this wave's 2 KiB slice of every tile
  // The LDS-DMA is issued from inline asm ON PURPOSE: hipcc treats a builtin LDS-DMA as a pending LDS write and parks an
  // s_waitcnt vmcnt(0) in front of the next ds_read, which here would stall every iteration on the tiles it has just requested.
  // An asm DMA is invisible to that bookkeeping; its completion is waited for by hand before the barrier that ends the
  // iteration, i.e. up to one iteration after its issue.  (M0 carries the LDS destination and is restored: hipcc owns it.)
  // piece i (0 / 1) of tile `tile` of K (which = 0) or V (which = 1) to LDS byte offset `dst_tile` (scalar: start of the buffer)
  auto dma_one = [&](auto which_c, auto i_c, int dst_tile, int tile) {
    constexpr int which = decltype(which_c)::value;
    constexpr int i = decltype(i_c)::value;
    const int64_t sn = which ? p.v_sn : p.k_sn;
    const char* base = which ? vbase_u : kbase_u;
    unsigned off = which ? (i ? voff1 : voff0) : (i ? koff1 : koff0);
    if (tile == nt_ - 1) {   // scalar branch, taken four times per workgroup: the last tile's rows past Nk re-read the last key
      off = last_off(i, sn); //   (their scores are masked); offsets relative to the (batch, head) base, recomputed here
    } else {
      base += (int64_t)tile * (128 * sn);                                   // scalar: 64 rows x stride x 2 bytes per tile
    }
    unsigned keep;
    const unsigned dst = lds_wave + (unsigned)dst_tile + 1024u * i;   // scalar
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(off), "s"(base), "s"(dst) : "memory");
  };
  using I0 = std::integral_constant<int, 0>;
  using I1 = std::integral_constant<int, 1>;
  auto dma = [&](auto which_c, int dst_tile, int tile) {
    dma_one(which_c, I0{}, dst_tile, tile);
    dma_one(which_c, I1{}, dst_tile, tile);
  };
  auto dma_wait_and_barrier = [&]() {   // every wave's outstanding LDS-DMA has landed, then the workgroup meets
    asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
  };
  using KOP = std::integral_constant<int, 0>;
  using VOP = std::integral_constant<int, 1>;

  // ---- per-lane LDS read offsets (the image of attn_fwd.hip::tile_off); set again from lane_now() after the loop so
  // that the loop's copies do not stay live through the register-hungry tail (which made hipcc spill them EVERYWHERE) ----
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
        v_off[half][d] = V_REGION + slot * TILE + 256 * (4 * hh + 8 * half + q4) + 8 * (p4 & 1) + 64 * (d ^ q4) + 16 * ((2 * g1 + (p4 >> 1)) ^ (hh + 2 * half));
  };
  set_read_offsets(lane, 2);   // (V slot 2: iteration 0 rotates the offsets to slot 0)

  auto read_k = [&](const lds_u8p* kb, int i) -> bf16x8 {   // fragment of score MFMA i: k-step i >> 1, key block i & 1
    return *reinterpret_cast<const AS3P bf16x8*>(kb + (i & 1) * 32 * 256 + k_off[i >> 1]);
  };
  // Q fragments (B operand of the score MFMAs): lane holds Q[q0 + r][16 ks + 8 h .. + 8], resident for the whole sweep
  bf16x8 qf[8];
  {
    int64_t qrow = q0 + r;
    if (qrow > p.Nq - 1) qrow = p.Nq - 1;
    const bf16_t* qp = p.q + b * p.q_sb + qrow * p.q_sn + (int64_t)head * p.q_sh + 8 * h;
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) qf[ks] = *reinterpret_cast<const bf16x8*>(qp + 16 * ks);
  }
  // fragment of PV MFMA j (k-step j >> 2, dim block j & 3) of the V tile the offsets currently point at (v_slot)
  auto read_v = [&](int j) -> bf16x8 {
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((AS3P s16x4*)(lds + 4096 * (j >> 2) + v_off[0][j & 3]));
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((AS3P s16x4*)(lds + 4096 * (j >> 2) + v_off[1][j & 3]));
    return __builtin_bit_cast(bf16x8, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
  };

  f32x16 oacc[4];
#pragma unroll
  for (int d = 0; d < 4; ++d)
#pragma unroll
    for (int e = 0; e < 16; ++e) oacc[d][e] = 0.f;
  float m_run = 0.f, l_run = 0.f;
  f32x16 minit;   // -m_run in every element: the C operand of both score chains' first MFMAs (changes only on a rescale)
#pragma unroll
  for (int e = 0; e < 16; ++e) minit[e] = 0.f;
  f32x16 sa0, sa1, sb0, sb1;   // score sets A and B

  const int nt = (int)((p.Nk + 63) / 64);
  const bool ragged = (p.Nk & 63) != 0;

  // scores of the (possibly ragged) last tile past Nk -> -inf (before their row max)
  auto mask_last = [&](f32x16& s0, f32x16& s1) {
    const int valid = (int)(p.Nk - (int64_t)(nt - 1) * 64);
    const int hh_ = lane_now() >> 5;
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int key = (e & 3) + 8 * (e >> 2) + 4 * hh_;
      if (key >= valid) s0[e] = -INFINITY;
      if (key + 32 >= valid) s1[e] = -INFINITY;
    }
  };
  // row max of a score tile relative to the running max, and the (rare) rescale it may trigger
  auto settle = [&](f32x16& s0, f32x16& s1, float mx, bool first) {   // mx: row max over both lane halves
    if (__builtin_amdgcn_ballot_w64(mx > PIPE_RESCALE_THR) != 0ull || first) {
      const float d = first ? mx : fmaxf(mx, 0.f);
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
  };

  // fragments that live across phase / iteration boundaries
  // LDS fragments are requested PD MFMAs (K, V) / two k-steps (Q) ahead of their use: under load an LDS read takes well over
  // two MFMA gaps to return (the first version, two gaps ahead, spent ~60 cycles per MFMA and wave in the phases)
  constexpr int PD = 2, RING = PD + 1;
  bf16x8 kfr[RING], vfr[RING];
  int v_slot = 2;   // V ring slot the transposed-read offsets point at (tile t lives in slot t % 3; iteration 0 rotates 2 -> 0)
  // rotate the eight V read offsets to the next tile's slot: +16 KiB, or -32 KiB when wrapping (a scalar operand)
  auto next_v_slot = [&]() -> int {
    v_slot = (v_slot == 2) ? 0 : v_slot + 1;
    return (v_slot == 0) ? -2 * TILE : TILE;
  };

  // ---- prologue: Q, K(0), V(0), K(1), V(1) requested; S(0) computed plainly and settled; then K(2) into K(0)'s buffer ----
  dma(KOP{}, 0, 0);                       // (the launcher guarantees nt >= 6)
  dma(VOP{}, V_REGION, 0);
  dma(KOP{}, TILE, 1);
  dma(VOP{}, V_REGION + TILE, 1);
  dma_wait_and_barrier();
  {
    const lds_u8p* kb = lds;
#pragma unroll
    for (int e = 0; e < 16; ++e) { sa0[e] = 0.f; sa1[e] = 0.f; }
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const bf16x8 a = read_k(kb, i);
      const bf16x8 qq = qf[i >> 1];
      if (i & 1) sa1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, qq, sa1, 0, 0, 0);
      else sa0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, qq, sa0, 0, 0, 0);
    }
    float mxa = sa0[0], mxb = sa1[0];
#pragma unroll
    for (int e = 1; e < 16; ++e) { mxa = fmaxf(mxa, sa0[e]); mxb = fmaxf(mxb, sa1[e]); }
    settle(sa0, sa1, pipe_half_max(fmaxf(mxa, mxb)), true);
  }
  __syncthreads();                        // every wave has read K(0)
  dma(KOP{}, 0, 2);                       // K(2) -> K buffer 0; waited for at the barrier of iteration 0
#pragma unroll
  for (int i = 0; i < PD; ++i) kfr[i] = read_k(lds + TILE, i);   // first fragments of K(1) and of Q: what an iteration expects

  // ---- one pipelined iteration.  PAR = t & 1: K(t+1) sits in K buffer PAR ^ 1, K(t+2) in buffer PAR, K(t+3) is requested
  // into buffer PAR ^ 1 after the barrier; V(t) in slot t % 3, V(t+2) requested into slot (t + 2) % 3.
  // (c0, c1) hold S(t), settled; (n0, n1) receive S(t+1).  STEADY: tiles up to t+3 exist and are full (no run-time checks).
  auto iteration = [&](const int t, auto par_c, f32x16& c0, f32x16& c1, f32x16& n0, f32x16& n1) {
    constexpr int PAR = decltype(par_c)::value;
    const lds_u8p* kb = lds + (PAR ^ 1) * TILE;      // K(t+1)
    const lds_u8p* kb_next = lds + PAR * TILE;        // K(t+2)
    // edge handling by scalar conditions (a few SALU instructions per iteration; no second, register-hungry loop body)
    const bool has_k3 = t + 3 < nt;
    const bool has_v2 = t + 2 < nt;
    const int v_delta = next_v_slot();                // v_slot == t % 3 from here on
    const int v_dst = V_REGION + ((v_slot == 0) ? 2 : v_slot - 1) * TILE;   // slot (t + 2) % 3
    phase_prio(1);
    float psum = 0.f;
    float ex[32];      // P(t) in fp32, element order of SC
    unsigned pw[16];   // P(t) as packed bf16 pairs: word m = elements (2m, 2m + 1)
    SCHED_FENCE();
    // ---------------- phase 1: 16 score MFMAs of tile t+1; exp2 / sums / packing of elements 0..23 of tile t ----------------
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      if (i + PD < 16) kfr[(i + PD) % RING] = read_k(kb, i + PD);
      if (i >= 16 - PD) vfr[i - (16 - PD)] = read_v(i - (16 - PD));   // first fragments of V(t) (landed since the last barrier)
      if (i == 0) {
        // both chains' first MFMAs in ONE statement, D != C (hipcc would pick the tied form and copy 16 registers per chain):
        // the resident -m_run tuple is read as C and survives
        asm volatile("v_mfma_f32_32x32x16_bf16 %0, %3, %5, %2\n\tv_mfma_f32_32x32x16_bf16 %1, %4, %5, %2"
                     : "=&v"(n0), "=&v"(n1) : "v"(minit), "v"(kfr[0]), "v"(kfr[1]), "v"(qf[0]));
      } else if (i == 1) {
        // (MFMA 1 was issued with MFMA 0; this gap only prefetches and carries its share of the vector work)
      } else if (i & 1) n1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kfr[i % RING], qf[i >> 1], n1, 0, 0, 0);
      else n0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kfr[i % RING], qf[i >> 1], n0, 0, 0, 0);
      SCHED_FENCE();
      // gaps 5..12: one of the eight V read offsets moves to tile t's slot (no V read is in flight between gap 0 and gap 14)
      if (i >= 5 && i < 13) asm volatile("v_add_u32 %0, %1, %0" : "+v"(v_off[(i - 5) >> 2][(i - 5) & 3]) : "s"(v_delta));
      // exps of this gap: elements [e_lo, e_hi); the sums and packs trail one gap behind
      const int e_lo = (3 * i + 1) / 2, e_hi = (3 * (i + 1) + 1) / 2;
      const int a_lo = i ? (3 * (i - 1) + 1) / 2 : 0, a_hi = i ? e_lo : 0;
#pragma unroll
      for (int j = 0; j < 24; ++j)
        if (j >= e_lo && j < e_hi) ex[j] = g_exp2(SC(c0, c1, j));
#pragma unroll
      for (int j = 0; j < 24; ++j)
        if (j >= a_lo && j < a_hi) {
          psum = (j == 0) ? ex[0] : g_add(psum, ex[j]);
          if (j & 1) pw[j >> 1] = g_pack(ex[j - 1], ex[j]);
        }
      SCHED_FENCE();
    }
    if (t + 1 == nt - 1 && ragged) mask_last(n0, n1);   // scalar branch, taken once
    // the one barrier: K(t+2) and V(t+1) (requested in phase 2 of iteration t-1) are in LDS for every wave; every wave has
    // finished reading K(t+1) (phase 1) and V(t-1) (phase 2 of iteration t-1)
    dma_wait_and_barrier();
    phase_prio(2);
    // ---------------- phase 2: 16 PV MFMAs of tile t; the rest of P(t); row max of S(t+1); next requests and fragments ------
    float mxa = 0.f, mxb = 0.f, mx = 0.f;
    SCHED_FENCE();
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      if (j + PD < 16) vfr[(j + PD) % RING] = read_v(j + PD);
      if (j >= 16 - PD) kfr[j - (16 - PD)] = read_k(kb_next, j - (16 - PD));   // first fragments of K(t+2) (unused after the last one)
      const int kk = j >> 2;
      const u32x4 pbw = {pw[4 * kk], pw[4 * kk + 1], pw[4 * kk + 2], pw[4 * kk + 3]};
      oacc[j & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vfr[j % RING], __builtin_bit_cast(bf16x8, pbw), oacc[j & 3], 0, 0, 0);
      SCHED_FENCE();
      // the four LDS-DMA pieces of this iteration: K(t+3) into K(t+1)'s buffer, V(t+2) into V(t-1)'s slot (both free since
      // the barrier above); they are waited for at the next barrier, a whole iteration away
      if (j == 0 && has_k3) dma_one(KOP{}, I0{}, (PAR ^ 1) * TILE, t + 3);
      if (j == 1 && has_k3) dma_one(KOP{}, I1{}, (PAR ^ 1) * TILE, t + 3);
      if (j == 2 && has_v2) dma_one(VOP{}, I0{}, v_dst, t + 2);
      if (j == 3 && has_v2) dma_one(VOP{}, I1{}, v_dst, t + 2);
      if (j == 0) {   // element 23 (exp'ed in the last gap of phase 1)
        psum = g_add(psum, ex[23]);
        pw[11] = g_pack(ex[22], ex[23]);
      }
      if (j < 8) {    // quarter 3: one exp per gap, sum / pack one gap later
        ex[24 + j] = g_exp2(SC(c0, c1, 24 + j));
        if (j > 0) {
          psum = g_add(psum, ex[24 + j - 1]);
          if ((j - 1) & 1) pw[12 + ((j - 1) >> 1)] = g_pack(ex[24 + j - 2], ex[24 + j - 1]);
        }
      } else if (j == 8) {
        psum = g_add(psum, ex[31]);
        pw[15] = g_pack(ex[30], ex[31]);
        l_run = g_add(l_run, psum);
      }
      // row max of S(t+1): chain a over n0[0..15] and n1[0] in gaps 2..9, chain b over n1[1..15] in gaps 3..9
      if (j == 2) mxa = g_max3(n0[0], n0[1], n1[0]);
      if (j > 2 && j < 10) mxa = g_max3(mxa, n0[2 * (j - 2)], n0[2 * (j - 2) + 1]);
      if (j == 3) mxb = g_max3(n1[1], n1[2], n1[3]);
      if (j > 3 && j < 10) mxb = g_max3(mxb, n1[2 * (j - 2)], n1[2 * (j - 2) + 1]);
      if (j == 10) mx = pipe_half_max(g_max3(mxa, mxb, mxb));   // + the partner half's keys (one v_permlane32_swap)
      SCHED_FENCE();
    }
    settle(n0, n1, mx, false);
  };

  // last tile: nothing left to overlap with; (c0, c1) hold S(nt - 1), settled; its V tile landed before the last barrier
  auto final_tile = [&](f32x16& c0, f32x16& c1) {
    const int v_delta = next_v_slot();
#pragma unroll
    for (int i = 0; i < 8; ++i) v_off[i >> 2][i & 3] += v_delta;
    float psum = 0.f;
    float ex[32];
#pragma unroll
    for (int j = 0; j < 32; ++j) {
      ex[j] = __builtin_amdgcn_exp2f(SC(c0, c1, j));
      psum += ex[j];
    }
    l_run += psum;
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      const int kk = j >> 2;
      const u32x4 pbw = {pack_bf16x2(ex[8 * kk], ex[8 * kk + 1]), pack_bf16x2(ex[8 * kk + 2], ex[8 * kk + 3]),
                         pack_bf16x2(ex[8 * kk + 4], ex[8 * kk + 5]), pack_bf16x2(ex[8 * kk + 6], ex[8 * kk + 7])};
      oacc[j & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(read_v(j), __builtin_bit_cast(bf16x8, pbw), oacc[j & 3], 0, 0, 0);
    }
  };

  {
    using P0 = std::integral_constant<int, 0>;
    using P1 = std::integral_constant<int, 1>;
    int t = 0;
    for (; t + 1 <= nt - 2; t += 2) {
      iteration(t, P0{}, sa0, sa1, sb0, sb1);
      iteration(t + 1, P1{}, sb0, sb1, sa0, sa1);
    }
    if (t == nt - 2) iteration(t, P0{}, sa0, sa1, sb0, sb1);
    set_read_offsets(lane_now(), v_slot);   // (fresh copies for the last tile: see lane_now)
    if ((nt - 1) & 1) final_tile(sb0, sb1);
    else final_tile(sa0, sa1);
  }

  // ---- epilogue ----
  const float l_tot = pipe_half_sum(l_run);
  const float inv = 1.0f / l_tot;
  const int lane_l = lane_now();
  const int r_l = lane_l & 31, h_l = lane_l >> 5;
  const int64_t qrow = q0 + r_l;
  if (qrow < p.Nq) {
    bf16_t* op = p.o + b * p.o_sb + qrow * p.o_sn + (int64_t)head * p.o_sh;
#pragma unroll
    for (int d = 0; d < 4; ++d)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        u16x4 pk;
#pragma unroll
        for (int e = 0; e < 4; ++e) pk[e] = f2bf(oacc[d][4 * i + e] * inv);
        *reinterpret_cast<u16x4*>(op + 32 * d + 8 * i + 4 * h_l) = pk;
      }
    if (p.lse && h_l == 0) p.lse[(b * p.H + head) * p.Nq + qrow] = m_run * p.scale + __logf(l_tot);
  }
}

// called by lcv_attn_fwd (attn_fwd.hip) for unit-scale self-attention with Nk >= 256
int attn_fwd_pipe_launch(const void* q, const void* k, const void* v, void* o, float* lse, int64_t B, int64_t H, int64_t Nq,
                         int64_t Nk, int64_t q_sb, int64_t q_sn, int64_t q_sh, int64_t k_sb, int64_t k_sn, int64_t k_sh,
                         int64_t v_sb, int64_t v_sn, int64_t v_sh, int64_t o_sb, int64_t o_sn, int64_t o_sh, float scale,
                         int xcd_ok, hipStream_t s) {
  AttnFwdPipeParams p;
  p.q = (const bf16_t*)q; p.k = (const bf16_t*)k; p.v = (const bf16_t*)v; p.o = (bf16_t*)o; p.lse = lse;
  p.Nq = Nq; p.Nk = Nk; p.H = (int)H;
  p.q_sb = q_sb; p.q_sn = q_sn; p.q_sh = q_sh; p.k_sb = k_sb; p.k_sn = k_sn; p.k_sh = k_sh;
  p.v_sb = v_sb; p.v_sn = v_sn; p.v_sh = v_sh; p.o_sb = o_sb; p.o_sn = o_sn; p.o_sh = o_sh;
  p.scale = scale;
  const unsigned gx = (unsigned)((Nq + 255) / 256);
  p.gx = (int)gx;
  p.xcd_remap = (xcd_ok && (B * H) % 8 == 0 && gx >= 8) ? 1 : 0;
  p.prio_mode = 0;
  const size_t lds = 5 * 64 * 256;   // K x2, V x3
  // (function-local static: initialised once, thread-safe)
  static const bool attr_ok = !(hipFuncSetAttribute((const void*)attn_fwd_pipe_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess);
  if (!attr_ok) {
      lcv_set_error("attn_fwd: cannot raise dynamic LDS");
      return LCV_EDEVICE;
  }
  const dim3 grid = p.xcd_remap ? dim3(gx * (unsigned)(H * B)) : dim3(gx, (unsigned)H, (unsigned)B);
  hipLaunchKernelGGL(attn_fwd_pipe_kernel, grid, dim3(512), lds, s, p);
  LCV_LAUNCH_CHECK("attn_fwd_pipe");
  return LCV_OK;
}
