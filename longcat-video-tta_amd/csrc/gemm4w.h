// 256 x 256 output tile on FOUR waves (one per SIMD), each owning 128 x 128; persistent workgroups that treat the K stages of
// all their output tiles as ONE stream.  Included by gemm.hip (uses GemmParams, gemm_tile_coords, gemm16_epilogue).
//
// Why (PMC, profiles/r01_pmc_mfma_busy.md): the 8-wave 8-phase kernel keeps the matrix pipe 0.62 busy at ~1.65 GHz, the vendor
// kernel for the same shapes 0.85 at 1.50 GHz.  The 8-wave layout reads every fragment twice (128 x 64 wave tiles: 192 KiB of
// ds_read + 64 KiB of LDS-DMA per 64-deep K tile = the whole LDS bandwidth of a CU) and pays two barriers per 16 MFMAs.  Here:
//   * a wave's 128 x 128 output = 8 x 8 MFMA tiles of 16x16x32 = 256 accumulator registers: the AGPR half of the 512-entry file
//     ("+a" on an asm MFMA; ONE definition of each tuple per tile and no branch between a tile's first and last MFMA -
//     with a second definition, or a branch in the accumulators' path, hipcc parks tuples in spare VGPRs and scratch);
//   * every fragment is read from LDS once per 64 MFMAs (128 KiB of ds_read per 64-deep K tile instead of 192);
//   * a "stage" is 32 deep: 512 rows x 64 B = 32 KiB (A rows 0..255 | W rows 0..255), four stage slots in LDS.  64-byte rows
//     make a 16x16x32 fragment one contiguous KiB (conflict-free ds_read_b128 without a swizzle) and an LDS-DMA instruction
//     16 rows x 64 B;
//   * the wave software-pipelines itself, everything from asm volatile statements (so the order below is the issue order):
//         phase p:  64 MFMAs on the fragments of stage p   (in registers since phase p-1)
//                   16 ds_read_b128: fragments of stage p+1          (spread over the first 7/8 of the phase)
//                    8 LDS-DMA: this wave's share of stage p+3        (one behind every 8th MFMA)
//     so the vector-memory path sees one instruction per wave every 128 cycles all the time (its 64 B/clk are half used;
//     issuing a K tile's 16 DMA back to back behind a barrier cost 25 %: the waves queue on the address path), a stage has
//     1-2 phases (1-2 thousand cycles) to land, and the LDS pipe carries 96 of its 128 B/clk;
//   * one barrier per phase, between two MFMA clusters:  s_waitcnt vmcnt(8) [my part of stage p+1 has landed; stage p+2 may
//     still be in flight]; s_barrier [everybody's has; everybody finished reading stage p-1, whose slot stage p+3 overwrites];
//   * the stream does not stop at an output-tile boundary: the DMA runs three stages ahead into the NEXT tile of the workgroup
//     and the last phase reads its first fragments, so the epilogue has the next tile's data in LDS and registers behind it.
// Past the end of the workgroup's stream the DMA re-fetches the last stage and the reads return fragments nobody uses (three
// surplus stages per workgroup) - no branch in the loop.
//
// G4_LAB_* macros: scratch/gemm_lab/g4_main.cpp only (wrong results on purpose: what does each ingredient of the loop cost?).
#pragma once

typedef __attribute__((address_space(3))) unsigned char g4_lds_u8;

template <int B, int E, class F>
__device__ __forceinline__ void g4_static_for(F&& f) {
  if constexpr (B < E) {
    f(std::integral_constant<int, B>{});
    g4_static_for<B + 1, E>(f);
  }
}

__device__ __forceinline__ void g4_mfma(f32x4v& acc, const bf16x8& w, const bf16x8& a) {
  asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(acc) : "v"(w), "v"(a));
}
template <int IMM>
__device__ __forceinline__ void g4_read(bf16x8& f, unsigned addr) {
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(f) : "v"(addr), "n"(IMM));
}
__device__ __forceinline__ void g4_set_m0(unsigned dst) { asm volatile("s_mov_b32 m0, %0" ::"s"(dst)); }
// (no immediate offset: the hardware adds it to the LDS address as well as to the global one)
__device__ __forceinline__ void g4_dma(unsigned off, const char* base) {
  asm volatile("global_load_lds_dwordx4 %0, %1" ::"v"(off), "s"(base) : "memory");
}
// NR reads spread over the first 7/8 of a phase of NG MFMAs: gap g carries read t iff t = floor(NR (g + 1) / span) - 1 is new
__host__ __device__ constexpr int g4_read_slot(int g, int NR, int NG) {
  const int span = NG * 7 / 8;
  return (g < span && NR * (g + 1) / span != NR * g / span) ? NR * (g + 1) / span - 1 : -1;
}

// NW = 4: one wave per SIMD, 128 x 128 per wave (the design above).  NW = 8: two waves per SIMD, 128 x 64 per wave - the same
// self-pipelined stream per wave, half the MFMAs / DMA pieces per wave and phase (32 / 4) and 12 fragment reads; the partner wave
// fills the issue slots a DMA piece or the epilogue takes, at the price of reading every A fragment twice per workgroup.
// (the interior-tile epilogue g4_fast_epilogue lives in gemm_fast_epilogue.h: the 8-phase kernel and the convolutions use it too)
template <int EPI, int NW>
__global__ __launch_bounds__(NW * 64) void gemm4w_nt_kernel(const GemmParams p) {
  constexpr unsigned SLOT = 32768u, W_OFF = 16384u, BUF = 65536u;
  constexpr int TN = NW == 4 ? 8 : 4;          // 16-column MFMA tiles per wave
  constexpr int NG = 8 * TN;                   // MFMAs per phase
  constexpr int NR = 8 + TN;                   // fragment reads per phase
  constexpr int NP = 16 / NW;                  // DMA pieces per wave, stage and operand
  constexpr int ROWS = 256 / NW;               // rows of A (and of W) a wave stages
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = NW == 4 ? wave >> 1 : wave >> 2, wc = NW == 4 ? wave & 1 : wave & 3;
  const int r16 = lane & 15, q = lane >> 4;
  const int vid_end = p.vid_begin + p.vid_count;
  const int nk1 = p.nk1, nk = p.nk1 + p.nk2;
  const int stride = (int)gridDim.x;
  const unsigned lds0 = (unsigned)(uintptr_t)(g4_lds_u8*)smem;
  const unsigned lds_wave = lds0 + (unsigned)wave * (unsigned)(ROWS * 64);   // this wave's 64 DMA rows of the A part of slot 0 (W part at + W_OFF)

  // ---- staging cursor: the (tile, K tile) whose stages the next DMA blocks fetch (half 0 then half 1 of a K tile: the phases
  // alternate in step with it).  A wave fills rows 64 w .. 64 w + 63 of the A and of the W part of a slot with 4 + 4 instructions
  // of 16 rows x 64 B; source = scalar base (operand + K offset + 64 B for k-half 1) + per-lane 32-bit offset ----
  int st_vid = p.vid_begin + (int)blockIdx.x, st_kt = 0;
  unsigned voa[NP], vow[NP];
  const char* st_a = nullptr;
  const char* st_w = nullptr;
  auto stage_setup = [&]() {
    int tm, tn;
    gemm_tile_coords(p, st_vid, tm, tn);
    const bool lora = st_kt >= nk1;   // the rank-r pair (a2, w2) supplies the last nk2 K tiles
    const unsigned lda_b = (unsigned)(lora ? p.lda2 : p.lda) * 2u, ldw_b = (unsigned)(lora ? p.ldw2 : p.ldw) * 2u;
    st_a = (const char*)(lora ? p.a2 : p.a);
    st_w = (const char*)(lora ? p.w2 : p.w);
#pragma unroll
    for (int t = 0; t < NP; ++t) {
      const int row = ROWS * wave + 16 * t + (lane >> 2);
      int64_t g = (int64_t)tm * 256 + row;
      voa[t] = (unsigned)(g > p.M - 1 ? p.M - 1 : g) * lda_b + (unsigned)(lane & 3) * 16u;
      g = (int64_t)tn * 256 + row;
      vow[t] = (unsigned)(g > p.N - 1 ? p.N - 1 : g) * ldw_b + (unsigned)(lane & 3) * 16u;
    }
  };
  auto stage_k_bytes = [&]() -> int64_t { return (int64_t)(st_kt >= nk1 ? st_kt - nk1 : st_kt) * 128; };
  auto stage_next_k_tile = [&]() {   // past the last K tile of the stream: stay on it
    if (st_kt + 1 == nk) {
      if (st_vid + stride < vid_end) {
        st_kt = 0;
        st_vid += stride;
        stage_setup();
      }
    } else {
      ++st_kt;
      if (st_kt == nk1) stage_setup();
    }
  };

  // ---- fragment read addresses: lane part + wave part, slot halves by the immediate, K-tile buffer by an XOR per K tile ----
  const unsigned lane_part = (unsigned)(r16 * 64 + q * 16);
  unsigned a_rd0 = lds0 + (unsigned)wr * 8192u + lane_part;            // next read of set 0 (k-half 0): buffer 0
  unsigned a_rd1 = a_rd0;                                             // next read of set 1 (k-half 1, + SLOT by the immediate)
  unsigned w_rd0 = lds0 + W_OFF + (unsigned)wc * (unsigned)(TN * 1024) + lane_part;
  unsigned w_rd1 = w_rd0;

  f32x4v acc[8][TN];
  bf16x8 fa[2][8], fw[2][TN];   // [set][i | j]

  auto read_frag = [&](auto set_c, auto t_c) {   // W fragments first: the first MFMA row of a phase needs all of them
    constexpr int S = decltype(set_c)::value, t = decltype(t_c)::value;
    constexpr int half_off = S ? (int)SLOT : 0;
    if constexpr (t < TN) g4_read<half_off + t * 1024>(fw[S][t], S ? w_rd1 : w_rd0);
    else g4_read<half_off + (t - TN) * 1024>(fa[S][t - TN], S ? a_rd1 : a_rd0);
  };
  // the 8 DMA of one stage (k-half H of the cursor's K tile) into the slot at LDS byte address dst (this wave's rows of it)
  auto dma_one = [&](auto h_c, auto t_c, const char* ab, const char* wb) {
    constexpr int H = decltype(h_c)::value, t = decltype(t_c)::value;
    if constexpr (t < NP) g4_dma(voa[t], ab + H * 64);
    else g4_dma(vow[t - NP], wb + H * 64);
  };

  // One phase: 64 MFMAs on set S; the reads of set S^1; the DMA of k-half H of the cursor's K tile into `dst`.
  auto phase = [&](auto s_c, auto h_c, unsigned dst, const char* ab, const char* wb) {
    constexpr int S = decltype(s_c)::value;
#if defined(G4_LAB_NO_BARRIER)
    if constexpr (NW == 4) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
#else
    if constexpr (NW == 4) asm volatile("s_waitcnt vmcnt(8)\n\ts_barrier" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(4)\n\ts_barrier" ::: "memory");
#endif
    g4_static_for<0, NG>([&](auto g_c) {
      constexpr int g = decltype(g_c)::value;
      constexpr int i = g / TN, j = g % TN;
      constexpr bool is_dma = (g & 7) == 4;
      constexpr int t_dma = g >> 3, t_rd = g4_read_slot(g, NR, NG);
      if constexpr (is_dma) g4_set_m0(dst + (t_dma < NP ? 0u : W_OFF) + (unsigned)(t_dma % NP) * 1024u);
      g4_mfma(acc[i][j], fw[S][j], fa[S][i]);
#ifndef G4_LAB_NO_DMA
      if constexpr (is_dma) dma_one(h_c, std::integral_constant<int, t_dma>{}, ab, wb);
#endif
#ifndef G4_LAB_NO_READS
      if constexpr (t_rd >= 0) read_frag(std::integral_constant<int, (S ^ 1)>{}, std::integral_constant<int, (t_rd < 0 ? 0 : t_rd)>{});
#endif
    });
    if constexpr (S == 0) { a_rd1 ^= BUF; w_rd1 ^= BUF; }   // set 1 was read from this K tile's buffer; next time the other one
    else { a_rd0 ^= BUF; w_rd0 ^= BUF; }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  };
  using C0 = std::integral_constant<int, 0>;
  using C1 = std::integral_constant<int, 1>;

  // ---- prologue: stages 0, 1, 2 of the stream (nk >= 2: K tiles 0 and 1 of the first tile), fragments of stage 0 ----
  int vid = st_vid;
  stage_setup();
  unsigned dq = 0;   // stages issued so far; stage q lives in slot q & 3
  auto dma_stage_now = [&](auto h_c) {
    const char* ab = st_a + stage_k_bytes();
    const char* wb = st_w + stage_k_bytes();
    const unsigned dst = lds_wave + (dq & 3u) * SLOT;
    g4_static_for<0, 2 * NP>([&](auto t_c) {
      constexpr int t = decltype(t_c)::value;
      g4_set_m0(dst + (t < NP ? 0u : W_OFF) + (unsigned)(t % NP) * 1024u);
      asm volatile("s_nop 0");
      dma_one(h_c, t_c, ab, wb);
    });
    ++dq;
  };
  dma_stage_now(C0{});
  dma_stage_now(C1{});
  stage_next_k_tile();
  dma_stage_now(C0{});   // the cursor now stands on k-half 1 of K tile 1: what phase 0 issues
  if constexpr (NW == 4) asm volatile("s_waitcnt vmcnt(16)\n\ts_barrier" ::: "memory");
  else asm volatile("s_waitcnt vmcnt(8)\n\ts_barrier" ::: "memory");
  g4_static_for<0, NR>([&](auto t_c) { read_frag(C0{}, t_c); });
  a_rd0 ^= BUF;
  w_rd0 ^= BUF;
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");

  for (;;) {
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j) acc[i][j] = f32x4v{0.f, 0.f, 0.f, 0.f};
    asm volatile("s_nop 7" ::: "memory");
    // The previous epilogue's stores read their data registers asynchronously; hipcc would otherwise park an s_waitcnt vmcnt(0)
    // in front of the first fragment read that reuses one of them - INSIDE the K loop, where it would also wait, every K tile,
    // for the LDS-DMA issued a phase earlier (asm DMA is invisible to its bookkeeping).  A wait it can see, once per tile:
    __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0)
    for (int kt = 0; kt < nk; ++kt) {
      // phase 2s: MFMAs on k-half 0; reads k-half 1 of this K tile; DMA = k-half 1 of the cursor's K tile (stream position s + 1)
      phase(C0{}, C1{}, lds_wave + (dq & 3u) * SLOT, st_a + stage_k_bytes(), st_w + stage_k_bytes());
      ++dq;
      stage_next_k_tile();
      // phase 2s + 1: MFMAs on k-half 1; reads k-half 0 of the next position; DMA = k-half 0 of position s + 2
      phase(C1{}, C0{}, lds_wave + (dq & 3u) * SLOT, st_a + stage_k_bytes(), st_w + stage_k_bytes());
      ++dq;
    }
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");   // the last MFMAs retire before the compiler's accumulator reads
    {
      int tm, tn;
      gemm_tile_coords(p, vid, tm, tn);
#ifdef G4_LAB_TRIVIAL_EPI
      float* cp = (float*)p.c + ((int64_t)tm * 256 + wr * 128) * p.ldc + (int64_t)tn * 256 + wc * (16 * TN) + lane * 4;
#pragma unroll
      for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) *reinterpret_cast<f32x4v*>(cp + (i * TN + j) * 256) = acc[i][j];
#else
      const int64_t mw = (int64_t)tm * 256 + wr * 128, nw = (int64_t)tn * 256 + wc * (16 * TN);
      if (g4_fast_epilogue_ok<EPI, TN>(p, mw, nw)) g4_fast_epilogue<EPI, TN>(p, acc, mw, nw, r16, q);
      else gemm16_epilogue<8, TN, EPI, true>(p, acc, mw, nw, r16, q);
#endif
    }
    vid += stride;
    if (vid >= vid_end) break;
    // The next tile's first fragments were read by the last phase; reading them AGAIN here makes those registers dead across
    // the epilogue (hipcc otherwise keeps 48-64 of them live through it and spills the epilogue's own values to scratch:
    // 100-250 scratch accesses per tile, each behind a full vmcnt wait - 30-40 us per tile).  One LDS latency per tile.
    a_rd0 ^= BUF;
    w_rd0 ^= BUF;
    g4_static_for<0, NR>([&](auto t_c) { read_frag(C0{}, t_c); });
    a_rd0 ^= BUF;
    w_rd0 ^= BUF;
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the surplus DMA blocks land before the LDS allocation is released
}
