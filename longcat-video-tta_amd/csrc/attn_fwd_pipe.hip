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
//                  phase 2   O += V(t)^T P(t)^T       16 MFMA   ||  quarter 3 of P(t), then the row max of S(t+1)
//                  post      rare: rescale O, l and S(t+1) when the running max grew by more than 2^RESCALE_THR
//
// The score accumulators of tile t+1 start at -running_max (the C operand of the first MFMA of each chain), so
// P = exp2(accumulator) with no subtraction; two named score register sets (A / B) swap roles every iteration (loop unrolled by
// two, every LDS address a register + immediate).  LDS: K(t+1) and V(t) are read while K(t+2) and V(t+1) land (two K and two V
// buffers, 64 KiB, one barrier per iteration).  Each gap's instructions are pinned with sched_barrier(0): the order in this
// file IS the order in the binary.  Registers: two score sets (64) + O (64) + the resident -running_max tuple (16) leave no room
// for the 32 registers of Q fragments, so every wave's 32 Q rows sit in LDS (8 KiB per wave, the K image and the K read offsets)
// and are read once per k-step beside the two K fragments.
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
};

#define PIPE_RESCALE_THR 6.0f

// Diagnostic build only (scratch/attn_lab/build_pipe.sh defines LCV_ATTN_STAMPS; the product never does): s_memtime stamps of
// waves 0 and 4 of one block at five points of eight consecutive iterations, written to a buffer nothing else reads.
#ifdef LCV_ATTN_STAMPS
__device__ unsigned long long* g_pipe_dbg = nullptr;
__device__ int g_pipe_dbg_block = 0;
#define PIPE_STAMP(id)                                                                                      \
  if (dbg_on && t >= 200 && t < 208) {                                                                      \
    unsigned long long t_;                                                                                  \
    __builtin_amdgcn_sched_barrier(0);                                                                      \
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");                              \
    __builtin_amdgcn_sched_barrier(0);                                                                      \
    /* parked in the last 32 KiB of LDS (no vector-memory traffic, so the loop's own vmcnt waits see nothing of it) */ \
    if (lane == 0) *reinterpret_cast<AS3P unsigned long long*>(lds + 131072 + wave * 4096 + ((t - 200) * 8 + (id)) * 8) = t_; \
  }
extern "C" void attn_pipe_set_stamps(unsigned long long* buf, int block) {
  (void)hipMemcpyToSymbol(HIP_SYMBOL(g_pipe_dbg), &buf, sizeof(buf));
  (void)hipMemcpyToSymbol(HIP_SYMBOL(g_pipe_dbg_block), &block, sizeof(block));
}
#else
#define PIPE_STAMP(id)
#endif

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
  lds_u8p* lds = (lds_u8p*)smem;  // K buffer 0 | K buffer 1 | V buffer 0 | V buffer 1 | Q rows of wave 0 .. 7 (8 KiB each)

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
#ifdef LCV_ATTN_STAMPS
  const bool dbg_on = g_pipe_dbg != nullptr && (int)blockIdx.x == g_pipe_dbg_block && (wave == 0 || wave == 4);
#endif
  const bf16_t* kbase = p.k + b * p.k_sb + (int64_t)head * p.k_sh;
  const bf16_t* vbase = p.v + b * p.v_sb + (int64_t)head * p.v_sh;

  // ---- LDS-DMA roles: wave w fills rows 8 w .. 8 w + 7 of a tile with two 1-KiB instructions ----
  // Source address = scalar base of the tile (SGPR pair, advanced one tile per issue by scalar adds) + a per-lane 32-bit byte
  // offset that never changes (row 8 w + 4 i + (lane >> 4), swizzled 16-byte column): no vector address arithmetic in the loop.
  unsigned koff[2], voff[2];
  // `lane_late` is the lane id again, made opaque AFTER the steady-state loop: everything only the tail iterations and the
  // epilogue need (ragged-row offsets, key indices of the mask, output pointers) is computed from it and therefore cannot be
  // hoisted in front of the loop, where it would sit in ~50 registers for the whole sweep and push loop values into scratch
  int lane_late = lane;
  auto dma_row_of = [&](int ln, int i) { return 8 * wave + 4 * i + (ln >> 4); };
  auto dma_colb_of = [&](int ln, int i) {
    const int row = dma_row_of(ln, i);
    return 16 * ((ln & 15) ^ (((row & 3) << 2) | ((row >> 2) & 3)));
  };
  auto set_dma_offsets = [&](int ln) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      koff[i] = (unsigned)(dma_row_of(ln, i) * p.k_sn * 2 + dma_colb_of(ln, i));
      voff[i] = (unsigned)(dma_row_of(ln, i) * p.v_sn * 2 + dma_colb_of(ln, i));
    }
  };
  set_dma_offsets(lane);
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
  // piece i (0 / 1) of tile `tile` of K (which = 0) or V (which = 1) into buffer `buf`
  auto dma_one = [&](auto which_c, auto i_c, auto buf_c, int tile, bool full) {
    constexpr int which = decltype(which_c)::value;
    constexpr int i = decltype(i_c)::value;
    constexpr int dst_off = (2 * which + decltype(buf_c)::value) * TILE + 1024 * i;
    const int64_t sn = which ? p.v_sn : p.k_sn;
    const char* base = which ? vbase_u : kbase_u;
    unsigned off;
    if (full) {
      base += (int64_t)tile * (128 * sn);            // scalar: 64 rows x stride x 2 bytes per tile
      off = which ? voff[i] : koff[i];
    } else {  // ragged last tile: rows past Nk re-read the last key (their scores are masked)
      int64_t row = (int64_t)tile * 64 + dma_row_of(lane_late, i);
      if (row > p.Nk - 1) row = p.Nk - 1;
      off = (unsigned)(row * sn * 2 + dma_colb_of(lane_late, i));
    }
    unsigned keep;
    const unsigned lw = lds_wave;   // (a captured variable cannot be an asm operand of a generic lambda directly)
    asm volatile("s_mov_b32 %0, m0\n\ts_add_u32 m0, %3, %4\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(off), "s"(base), "s"(lw), "i"(dst_off) : "memory", "scc");
  };
  using I0 = std::integral_constant<int, 0>;
  using I1 = std::integral_constant<int, 1>;
  auto dma = [&](auto which_c, auto buf_c, int tile, bool full) {
    dma_one(which_c, I0{}, buf_c, tile, full);
    dma_one(which_c, I1{}, buf_c, tile, full);
  };
  // this wave's 32 Q rows -> its 8 KiB of the Q region, laid out FRAGMENT-major: plane ks (1 KiB) holds, for lane (r, h), the 16
  // bytes Q[q0 + r][16 ks + 8 h ..] at 32 r + 16 h, so a B-operand read is one base register + the immediate 1024 ks and is
  // conflict-free (consecutive lanes, consecutive 16-byte slots).  One LDS-DMA instruction fills one plane: lane L fetches row
  // L >> 1, 16-byte chunk 2 ks + (L & 1) (rows past Nq re-read the last row).
  auto dma_q = [&]() {
    const char* qbase_u = uniform_ptr(p.q + b * p.q_sb + (int64_t)head * p.q_sh);
    int64_t g = q0 + (lane >> 1);
    if (g > p.Nq - 1) g = p.Nq - 1;
    const unsigned row_off = (unsigned)(g * p.q_sn * 2 + 16 * (lane & 1));
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) {
      const unsigned off = row_off + 32u * ks;
      const unsigned dst = (unsigned)(uintptr_t)(lds + 4 * TILE) + (unsigned)wave * 8192u + 1024u * ks;
      unsigned keep;
      asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                   : "=&s"(keep) : "v"(off), "s"(qbase_u), "s"(dst) : "memory");
    }
  };
  auto dma_wait_and_barrier = [&]() {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
  };
  using KOP = std::integral_constant<int, 0>;
  using VOP = std::integral_constant<int, 1>;

  // ---- per-lane LDS read offsets (the image of attn_fwd.hip::tile_off); set again from `lane_late` after the steady loop so
  // that the loop's copies do not stay live through the register-hungry tail (which made hipcc spill them EVERYWHERE) ----
  int k_off[8];
  int v_off[2][4];
  const lds_u8p* qlane;
  auto set_read_offsets = [&](int ln) {
    const int rr = ln & 31, hh = ln >> 5;
    const int kfz = ((rr & 3) << 2) | ((rr >> 2) & 3);
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) k_off[ks] = 256 * rr + 16 * ((2 * ks + hh) ^ kfz);
    const int q4 = (ln >> 2) & 3, p4 = ln & 3, g1 = (ln >> 4) & 1;
#pragma unroll
    for (int half = 0; half < 2; ++half)
#pragma unroll
      for (int d = 0; d < 4; ++d)
        v_off[half][d] = 256 * (4 * hh + 8 * half + q4) + 8 * (p4 & 1) + 64 * (d ^ q4) + 16 * ((2 * g1 + (p4 >> 1)) ^ (hh + 2 * half));
    qlane = lds + 4 * TILE + wave * 8192 + 32 * rr + 16 * hh;
  };
  set_read_offsets(lane);

  auto read_k = [&](const lds_u8p* kb, int i) -> bf16x8 {   // fragment of score MFMA i: k-step i >> 1, key block i & 1
    return *reinterpret_cast<const AS3P bf16x8*>(kb + (i & 1) * 32 * 256 + k_off[i >> 1]);
  };
  auto read_q = [&](int ks) -> bf16x8 {   // B operand of k-step ks: Q[q0 + r][16 ks + 8 h ..]
    return *reinterpret_cast<const AS3P bf16x8*>(qlane + 1024 * ks);
  };
  auto read_v = [&](const lds_u8p* vb, int j) -> bf16x8 {   // fragment of PV MFMA j: k-step j >> 2, dim block j & 3
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((AS3P s16x4*)(vb + 4096 * (j >> 2) + v_off[0][j & 3]));
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((AS3P s16x4*)(vb + 4096 * (j >> 2) + v_off[1][j & 3]));
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
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int key = (e & 3) + 8 * (e >> 2) + 4 * (lane_late >> 5);
      if (key >= valid) s0[e] = -INFINITY;
      if (key + 32 >= valid) s1[e] = -INFINITY;
    }
  };
  // row max of a score tile relative to the running max, and the (rare) rescale it may trigger
  auto settle = [&](f32x16& s0, f32x16& s1, float mx, bool first) {
    mx = pipe_half_max(mx);
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

  // ---- prologue: K(0), V(0), K(1) in flight; S(0) computed plainly and settled ----
  dma_q();
  dma(KOP{}, I0{}, 0, true);              // (the launcher guarantees nt >= 4: tiles 0 and 1 are full)
  dma(VOP{}, I0{}, 0, true);
  dma(KOP{}, I1{}, 1, true);
  dma_wait_and_barrier();
  {
    const lds_u8p* kb = lds;
#pragma unroll
    for (int e = 0; e < 16; ++e) { sa0[e] = 0.f; sa1[e] = 0.f; }
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const bf16x8 a = read_k(kb, i);
      const bf16x8 qq = read_q(i >> 1);
      if (i & 1) sa1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, qq, sa1, 0, 0, 0);
      else sa0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, qq, sa0, 0, 0, 0);
    }
    float mxa = sa0[0], mxb = sa1[0];
#pragma unroll
    for (int e = 1; e < 16; ++e) { mxa = fmaxf(mxa, sa0[e]); mxb = fmaxf(mxb, sa1[e]); }
    settle(sa0, sa1, fmaxf(mxa, mxb), true);
  }

  // ---- one pipelined iteration.  PAR = t & 1 selects the buffers (K(t+1): PAR ^ 1, V(t): PAR; DMA targets K: PAR, V: PAR ^ 1);
  // (c0, c1) hold S(t), settled; (n0, n1) receive S(t+1).  STEADY: tiles t+1 and t+2 exist and are full (no runtime checks).
  auto iteration = [&](const int t, auto par_c, auto steady_c, f32x16& c0, f32x16& c1, f32x16& n0, f32x16& n1) {
    constexpr int PAR = decltype(par_c)::value;
    constexpr bool STEADY = decltype(steady_c)::value;
    const lds_u8p* kb = lds + (PAR ^ 1) * TILE;
    const lds_u8p* vb = lds + (2 + PAR) * TILE;
    // run-time edge handling of the tail iterations (folds away when STEADY)
    const bool has_k = STEADY || (t + 2 < nt);
    const bool k_full = STEADY || (t + 2 < nt - 1) || !ragged;
    const bool v_full = STEADY || (t + 1 < nt - 1) || !ragged;
    PIPE_STAMP(0)
    float psum = 0.f;
    float ex[32];      // P(t) in fp32, element order of SC
    unsigned pw[16];   // P(t) as packed bf16 pairs: word m = elements (2m, 2m + 1)
    bf16x8 kfr[3], qfr[2];
    kfr[0] = read_k(kb, 0);
    kfr[1] = read_k(kb, 1);
    qfr[0] = read_q(0);
    SCHED_FENCE();
    // ---------------- phase 1: 16 score MFMAs of tile t+1; exp2 / sums / packing of elements 0..23 of tile t ----------------
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      if (i + 2 < 16) kfr[(i + 2) % 3] = read_k(kb, i + 2);
      if (!(i & 1) && i + 2 < 16) qfr[((i >> 1) + 1) & 1] = read_q((i >> 1) + 1);   // next k-step's Q fragment
      if (i == 0) {
        // both chains' first MFMAs in ONE statement, D != C (hipcc would pick the tied form and copy 16 registers per chain):
        // the resident -m_run tuple is read as C and survives
        asm volatile("v_mfma_f32_32x32x16_bf16 %0, %3, %5, %2\n\tv_mfma_f32_32x32x16_bf16 %1, %4, %5, %2"
                     : "=&v"(n0), "=&v"(n1) : "v"(minit), "v"(kfr[0]), "v"(kfr[1]), "v"(qfr[0]));
      } else if (i == 1) {
        // (MFMA 1 was issued with MFMA 0; this gap only prefetches and carries its share of the vector work)
      } else if (i & 1) n1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kfr[i % 3], qfr[(i >> 1) & 1], n1, 0, 0, 0);
      else n0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kfr[i % 3], qfr[(i >> 1) & 1], n0, 0, 0, 0);
      SCHED_FENCE();
      // the four LDS-DMA pieces of this iteration (K(t+2) -> K buffer PAR, V(t+1) -> V buffer PAR ^ 1; both buffers were
      // last read in iteration t-1) go into gaps 1, 2, 3, 4: an issue costs ~60 cycles when nothing else is running
      using BK = std::integral_constant<int, PAR>;
      using BV = std::integral_constant<int, PAR ^ 1>;
      if (i == 1 && has_k) dma_one(KOP{}, I0{}, BK{}, t + 2, k_full);
      if (i == 2 && has_k) dma_one(KOP{}, I1{}, BK{}, t + 2, k_full);
      if (i == 3) dma_one(VOP{}, I0{}, BV{}, t + 1, v_full);
      if (i == 4) dma_one(VOP{}, I1{}, BV{}, t + 1, v_full);
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
    PIPE_STAMP(1)
    // ---------------- phase 2: 16 PV MFMAs of tile t; element 23's trailing work, quarter 3 of tile t, row max of tile t+1 ----
    if (!STEADY && t + 1 == nt - 1 && ragged) mask_last(n0, n1);
    bf16x8 vfr[3];
    vfr[0] = read_v(vb, 0);
    vfr[1] = read_v(vb, 1);
    float mxa = 0.f, mxb = 0.f;
    SCHED_FENCE();
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      if (j + 2 < 16) vfr[(j + 2) % 3] = read_v(vb, j + 2);
      const int kk = j >> 2;
      const u32x4 pbw = {pw[4 * kk], pw[4 * kk + 1], pw[4 * kk + 2], pw[4 * kk + 3]};
      oacc[j & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vfr[j % 3], __builtin_bit_cast(bf16x8, pbw), oacc[j & 3], 0, 0, 0);
      SCHED_FENCE();
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
        mxa = g_max3(n0[0], n0[1], n1[0]);
      } else {        // row max of S(t+1)
        const int e = 2 * (j - 8);
        mxa = g_max3(mxa, n0[e], n0[e + 1]);
        mxb = (j == 9) ? g_max3(n1[1], n1[e], n1[e + 1]) : g_max3(mxb, n1[e], n1[e + 1]);
      }
      SCHED_FENCE();
    }
    PIPE_STAMP(2)
    settle(n0, n1, fmaxf(mxa, mxb), false);
    PIPE_STAMP(3)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    PIPE_STAMP(4)
    __syncthreads();          // K(t+2) / V(t+1) landed for every wave; this iteration's LDS reads are done
    PIPE_STAMP(5)
  };

  // last tile: nothing left to overlap with; (c0, c1) hold S(nt - 1), settled
  auto final_tile = [&](auto par_c, f32x16& c0, f32x16& c1) {
    constexpr int PAR = decltype(par_c)::value;
    const lds_u8p* vb = lds + (2 + PAR) * TILE;
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
      oacc[j & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(read_v(vb, j), __builtin_bit_cast(bf16x8, pbw), oacc[j & 3], 0, 0, 0);
    }
  };

  {
    using P0 = std::integral_constant<int, 0>;
    using P1 = std::integral_constant<int, 1>;
    using Y = std::true_type;
    using N = std::false_type;
    int t = 0;
    for (; t + 1 <= nt - 4; t += 2) {            // iterations t and t+1: tiles up to t+3 <= nt-2 are full
      iteration(t, P0{}, Y{}, sa0, sa1, sb0, sb1);
      iteration(t + 1, P1{}, Y{}, sb0, sb1, sa0, sa1);
    }
    asm volatile("" : "+v"(lane_late));          // (see lane_late above)
    set_dma_offsets(lane_late);
    set_read_offsets(lane_late);
    for (; t <= nt - 2; ++t) {                    // at most three tail iterations with run-time edge handling
      if (t & 1) iteration(t, P1{}, N{}, sb0, sb1, sa0, sa1);
      else iteration(t, P0{}, N{}, sa0, sa1, sb0, sb1);
    }
    if ((nt - 1) & 1) final_tile(P1{}, sb0, sb1);
    else final_tile(P0{}, sa0, sa1);
  }

#ifdef LCV_ATTN_STAMPS
  if (dbg_on && lane == 0)
    for (int i = 0; i < 64; ++i)
      g_pipe_dbg[(wave ? 256 : 0) + i] = *reinterpret_cast<AS3P unsigned long long*>(lds + 131072 + wave * 4096 + i * 8);
#endif
  // ---- epilogue ----
  const float l_tot = pipe_half_sum(l_run);
  const float inv = 1.0f / l_tot;
  const int r_l = lane_late & 31, h_l = lane_late >> 5;
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
#ifdef LCV_ATTN_STAMPS
  const size_t lds = 163840;
#else
  const size_t lds = 4 * 64 * 256 + 8 * 8192;   // K x2, V x2, Q rows
#endif
  static bool attr_set = false;
  if (!attr_set) {
    if (hipFuncSetAttribute((const void*)attn_fwd_pipe_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) {
      lcv_set_error("attn_fwd: cannot raise dynamic LDS");
      return LCV_EDEVICE;
    }
    attr_set = true;
  }
  const dim3 grid = p.xcd_remap ? dim3(gx * (unsigned)(H * B)) : dim3(gx, (unsigned)H, (unsigned)B);
  hipLaunchKernelGGL(attn_fwd_pipe_kernel, grid, dim3(512), lds, s, p);
  LCV_LAUNCH_CHECK("attn_fwd_pipe");
  return LCV_OK;
}
