// 256 x 256 output tile on FOUR waves (one per SIMD, 128 x 128 each), 64-deep K tiles in 128-byte LDS rows, persistent workgroups
// that treat the K tiles of all their output tiles as ONE stream.  Included by gemm.hip (GemmParams, gemm_tile_coords,
// gemm16_epilogue); `LCV_GEMM_TILE=k`.  Round 4: what the counters of profiles/r04_gemm_ab.md say the round-2 four-wave kernel
// (`gemm4w.h`, now scratch/tried/) still paid for, against the vendor kernel of the same tile (hipBLASLt `..._MT256x256x64_MI16x16x1`, 0.80 matrix-pipe
// busy at 1.49 GHz vs 0.61 at 1.88):
//   * gemm4w stages 32-deep half K tiles in 64-byte rows (a fragment is then one contiguous KiB): every 128-byte line of A and W
//     is requested TWICE from L2, as two half lines at different times - TCP_TCC_READ_REQ 3.87e8 per launch against 1.92e8 for the
//     vendor kernel and for the 8-phase kernel.  Here a stage is a whole 64-deep K tile in 128-byte rows (one request per line),
//     kept conflict-free for ds_read_b128 by an XOR of the 16-byte chunk index with (row & 7), applied on the SOURCE side of the
//     LDS-DMA (lane l of a piece lands in chunk l & 7 of row l >> 3 and fetches chunk (l & 7) ^ (l >> 3));
//   * two K-tile buffers (128 KiB), each half of a buffer (W rows, A rows) refilled as soon as its last reader has passed: three
//     barriers per K tile, 16 LDS-DMA pieces per wave in two groups of 8 behind them, `vmcnt(16)` in front of the third - a
//     request has more than one K tile (~2 500 cycles) to land.  (The first form of this kernel had ONE barrier per K tile and
//     requested tile t + 2 in the second half of tile t for use at the end of the first half of t + 1: ~1 000 cycles of slack,
//     less than a loaded L2 miss - no faster than gemm4w; profiles/r04_gemm_ab.md.)  The slot table is at `k_tile` below;
//   * the epilogue stores 16 bytes per lane and instruction: the weight rows of every 32-column block are dealt to its two
//     16-row MFMA tiles so that a lane's values of tiles (2u, 2u + 1) are 8 consecutive columns (a permutation of the DMA's source
//     rows, free) - 32 store instructions per lane and tile instead of 64, each covering 64-byte row segments instead of 32.
// Kept from gemm4w: accumulators pinned to the AGPR half of the register file by "+a" asm MFMAs (one definition per
// tile, no branch between a tile's first and last MFMA), every fragment read from LDS once per 64 MFMAs, the stream runs on into
// the next output tile (its first K tiles are in LDS and its first fragments in registers behind the epilogue), surplus stages
// past the end of the stream re-fetch the last K tile.  Same MFMA, same K order as every other 16x16x32 kernel here: bit-identical
// results (tests/test_gpu_kernels.py uses that as the race screen).
#pragma once

typedef __attribute__((address_space(3))) unsigned char g4_lds_u8;

template <int B, int E, class F>
__device__ __forceinline__ void g4_static_for(F&& f) {
  if constexpr (B < E) {
    f(std::integral_constant<int, B>{});
    g4_static_for<B + 1, E>(f);
  }
}
// everything the K loop issues comes from asm volatile statements: the order in the source is the issue order
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

// ---- the ONE epilogue of this kernel.  With one wave per SIMD nothing runs beside it, and on this ISA loads, stores and LDS-DMA
// share ONE in-order counter (vmcnt): a wait for ANY load that was issued after a store also waits for that store to be
// acknowledged by memory.  The first form of this kernel (the interior-tile epilogue of the 8-phase kernel plus the generic one
// for edge tiles, chosen per tile) spilled, and every spill reload between two stores drained the store queue: 22 full drains per
// tile, 20 us of a 112 us tile at the qkv shape (profiles/r04_gemm_ab.md).  Hence: every load (bias, gate, residual rows) is
// issued BEFORE the first store of the rows it serves, so its wait counts only younger operations; there is no second code path
// (what a tile may need is decided on the host, `gemm4k_eligible`): rows past M are predicated off, a wave tile that straddles
// two latent frames keeps both frames' gate rows and picks per row.
// PAIRED accumulator layout: acc[i][j][e] = C[mw + 16 i + r16][nw + 32 (j >> 1) + 8 q + 4 (j & 1) + e]; the same expressions as
// gemm16_epilogue, operand for operand (bit-identical results), with 16-byte loads and stores.
template <int EPI>
static bool gemm4k_eligible(const GemmParams& p) {
  if (p.out_f32 || (p.N & 255) || (p.ldc & 7) || ((uintptr_t)p.c & 15) || p.nk1 + p.nk2 < 2) return false;
  if (p.bias && ((uintptr_t)p.bias & 15)) return false;
  if constexpr (EPI == LCV_EPI_NONE) return true;
  if constexpr (EPI == LCV_EPI_GATE_RESIDUAL)
    return ((uintptr_t)p.resid & 15) == 0 && (!p.gate || (((uintptr_t)p.gate & 15) == 0 && (p.mod_stride & 3) == 0 && p.rows_per_frame >= 128));
  if constexpr (EPI == LCV_EPI_SWIGLU) return !p.resid || (((uintptr_t)p.resid & 15) == 0 && (p.N & 7) == 0);
  return false;
}

template <int EPI>
__device__ __forceinline__ void g4k_epilogue(const GemmParams& p, f32x4v (&acc)[8][8], int64_t mw, int64_t nw, int r16, int q) {
  u16x8 b8[4];   // the lane's 4 x 8 bias values, packed
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    if (p.bias) b8[u] = *reinterpret_cast<const u16x8*>(p.bias + nw + 32 * u + 8 * q);
    else b8[u] = u16x8{0, 0, 0, 0, 0, 0, 0, 0};
  }
  const int64_t m_first = mw + r16;                       // this lane's row of block i is m_first + 16 i
  const int64_t m_last_ok = p.M - 1;
  if constexpr (EPI == LCV_EPI_NONE) {
    bf16_t* crow = (bf16_t*)p.c + m_first * p.ldc + nw + 8 * q;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const bool ok = m_first + 16 * i <= m_last_ok;
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        u16x8 o;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const float a = acc[i][2 * u + (e >> 2)][e & 3];
          o[e] = f2bf(p.bias ? a + bf2f(b8[u][e]) : a);
        }
        if (ok) *reinterpret_cast<u16x8*>(crow + 32 * u) = o;
      }
      crow += 16 * p.ldc;
      __builtin_amdgcn_sched_barrier(0);
    }
  } else if constexpr (EPI == LCV_EPI_GATE_RESIDUAL) {
    // gate rows: frame f0 = the tile's first row's; rows at or past `mb` belong to frame f0 + 1 (rows_per_frame >= 128: at most one
    // boundary inside 128 rows).  BOTH frames' gate values are kept (64 registers) and every lane picks per row block: no second
    // code path and no loop around the accumulator reads (hipcc hoists AGPR reads out of a loop: 256 live registers).
    const int64_t f0 = (mw <= m_last_ok ? mw : m_last_ok) / p.rows_per_frame;   // (a wave tile wholly past M must not index past the table)
    const int64_t mb = (f0 + 1) * p.rows_per_frame;
    const int64_t off0 = nw + 8 * q;
    f32x4 ga[4][2], gb[4][2];
    if (p.gate) {
      const float* grow = p.gate + f0 * p.mod_stride + off0;
      const float* grow_b = mb <= m_last_ok ? grow + p.mod_stride : grow;     // (no frame f0 + 1 when the matrix ends first)
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        ga[u][0] = *reinterpret_cast<const f32x4*>(grow + 32 * u); ga[u][1] = *reinterpret_cast<const f32x4*>(grow + 32 * u + 4);
        gb[u][0] = *reinterpret_cast<const f32x4*>(grow_b + 32 * u); gb[u][1] = *reinterpret_cast<const f32x4*>(grow_b + 32 * u + 4);
      }
    }
    // Row block by row block, pinned by compiler fences: the residual row of block i + 1 is requested BEFORE block i is stored, so
    // the wait for it counts only the four younger stores (no drain) and two residual rows are the most that is live (the first
    // form of this epilogue let hipcc hoist all 32 residual loads: 512 registers, and the K loop's own DMA sources were spilled -
    // a scratch reload and a full wait in front of every LDS-DMA piece, 320 TF/s).  A row past M reads row M - 1 (never stored).
#define G4K_FENCE() do { asm volatile("" ::: "memory"); __builtin_amdgcn_sched_barrier(0); } while (0)
    u16x8 rcur[4], rnxt[4], out[4];
    auto load_res = [&](int i, u16x8 (&r)[4]) {
      const int64_t m = m_first + 16 * i <= m_last_ok ? m_first + 16 * i : m_last_ok;
      const bf16_t* rrow = p.resid + m * p.ldc + off0;
#pragma unroll
      for (int u = 0; u < 4; ++u) r[u] = *reinterpret_cast<const u16x8*>(rrow + 32 * u);
    };
    load_res(0, rcur);
    G4K_FENCE();
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int64_t m = m_first + 16 * i;
      const bool hi = m >= mb;
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          float v = acc[i][2 * u + (e >> 2)][e & 3];
          if (p.bias) v += bf2f(b8[u][e]);
          const float g = p.gate ? (hi ? gb[u][e >> 2][e & 3] : ga[u][e >> 2][e & 3]) : 1.0f;
          out[u][e] = f2bf(bf2f(rcur[u][e]) + g * bfround(v));
        }
      G4K_FENCE();
      if (i < 7) load_res(i + 1, rnxt);
      G4K_FENCE();
      if (m <= m_last_ok) {
        bf16_t* crow = (bf16_t*)p.c + m * p.ldc + off0;
#pragma unroll
        for (int u = 0; u < 4; ++u) *reinterpret_cast<u16x8*>(crow + 32 * u) = out[u];
      }
      G4K_FENCE();
#pragma unroll
      for (int u = 0; u < 4; ++u) rcur[u] = rnxt[u];
    }
#undef G4K_FENCE
  } else if constexpr (EPI == LCV_EPI_SWIGLU) {
    // W rows interleaved [32 gate | 32 up] per 64 columns: 32-column blocks u = 2 b (gate) and 2 b + 1 (up) of the same 32 features
    bf16_t* crow = (bf16_t*)p.c + m_first * p.ldc + nw / 2 + 8 * q;
    bf16_t* arow = p.resid ? const_cast<bf16_t*>(p.resid) + m_first * p.N + nw + 8 * q : nullptr;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const bool ok = m_first + 16 * i <= m_last_ok;
#pragma unroll
      for (int b = 0; b < 2; ++b) {
        u16x8 o, og, ou;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const float bg = p.bias ? bf2f(b8[2 * b][e]) : 0.f;
          const float bu = p.bias ? bf2f(b8[2 * b + 1][e]) : 0.f;
          const float gvv = bfround(acc[i][4 * b + (e >> 2)][e & 3] + bg);
          const float uvv = bfround(acc[i][4 * b + 2 + (e >> 2)][e & 3] + bu);
          o[e] = f2bf(bfround(silu_f(gvv)) * uvv);
          og[e] = f2bf(gvv);
          ou[e] = f2bf(uvv);
        }
        if (ok) {
          *reinterpret_cast<u16x8*>(crow + 32 * b) = o;
          if (arow) {   // training: the pre-activation (gate | up) rows
            *reinterpret_cast<u16x8*>(arow + 64 * b) = og;
            *reinterpret_cast<u16x8*>(arow + 64 * b + 32) = ou;
          }
        }
      }
      crow += 16 * p.ldc;
      if (arow) arow += 16 * p.N;
      __builtin_amdgcn_sched_barrier(0);
    }
  }
}

template <int EPI>
__global__ __launch_bounds__(256) void gemm4k_nt_kernel(const GemmParams p) {
  constexpr unsigned BUF = 65536u, W_OFF = 32768u;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 1, wc = wave & 1;
  const int r16 = lane & 15, q = lane >> 4;
  const int vid_end = p.vid_begin + p.vid_count;
  const int nk1 = p.nk1, nk = p.nk1 + p.nk2;
  const int stride = (int)gridDim.x;
  const unsigned lds0 = (unsigned)(uintptr_t)(g4_lds_u8*)smem;
  const unsigned lds_wave = lds0 + (unsigned)wave * 8192u;   // this wave's 64 DMA rows of the A part of buffer 0 (W part at + W_OFF)

  // 16-byte chunk c of LDS row R holds logical chunk c ^ (R & 7).  (Lab, profiles/r04_gemm_ab.md: a key that keeps every lane
  // quad's 64 bytes ascending - 4 * bit 1 of the row, two-way conflicted reads - runs at the same speed; no swizzle: -10 %.)
  auto swz_key = [&](int row) -> int { return row & 7; };
  // ---- staging cursor: the (output tile, K tile) the next 16 DMA pieces fetch.  A wave fills rows 64 w .. 64 w + 63 of the A part
  // and of the W part with 8 + 8 pieces of 8 rows x 128 B; source = scalar base (operand + K offset) + per-lane 32-bit offset.
  // LDS row R of the W part holds weight row 32 (R >> 5) + 8 ((R & 15) >> 2) + (R & 3) + 4 ((R >> 4) & 1) of the tile (PAIRED).
  int st_vid = p.vid_begin + (int)blockIdx.x, st_kt = 0;
  unsigned voa[8], vow[8];
  const char* st_a = nullptr;
  const char* st_w = nullptr;
  auto stage_setup = [&]() {
    int tm, tn;
    gemm_tile_coords(p, st_vid, tm, tn);
    const bool lora = st_kt >= nk1;   // the rank-r pair (a2, w2) supplies the last nk2 K tiles
    const unsigned lda_b = (unsigned)(lora ? p.lda2 : p.lda) * 2u, ldw_b = (unsigned)(lora ? p.ldw2 : p.ldw) * 2u;
    st_a = (const char*)(lora ? p.a2 : p.a);
    st_w = (const char*)(lora ? p.w2 : p.w);
    const unsigned chunk = (unsigned)((lane & 7) ^ swz_key(lane >> 3)) * 16u;   // row & 7 == lane >> 3 for every piece
#pragma unroll
    for (int t = 0; t < 8; ++t) {
      const int R = 64 * wave + 8 * t + (lane >> 3);
      int64_t g = (int64_t)tm * 256 + R;
      voa[t] = (unsigned)(g > p.M - 1 ? p.M - 1 : g) * lda_b + chunk;
      g = (int64_t)tn * 256 + (R & ~31) + 8 * ((R & 15) >> 2) + (R & 3) + 4 * ((R >> 4) & 1);
      vow[t] = (unsigned)(g > p.N - 1 ? p.N - 1 : g) * ldw_b + chunk;
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

  // ---- fragment read addresses: fragment i of A = rows 128 wr + 16 i + r16, of W = LDS rows 128 wc + 16 j + r16; 16-byte chunk
  // (4 h + q) ^ (r16 & 7) of the row for k-half h; the immediate carries i (or j) * 2048; the K-tile buffer is an XOR per K tile ----
  const unsigned lane_h0 = (unsigned)(r16 * 128 + ((q ^ swz_key(r16)) * 16));
  const unsigned lane_h1 = (unsigned)(r16 * 128 + (((4 + q) ^ swz_key(r16)) * 16));
  unsigned a_rd0 = lds0 + (unsigned)wr * 16384u + lane_h0;             // next read of k-half 0: buffer 0
  unsigned a_rd1 = lds0 + (unsigned)wr * 16384u + lane_h1;             // next read of k-half 1: buffer 0
  unsigned w_rd0 = lds0 + W_OFF + (unsigned)wc * 16384u + lane_h0;
  unsigned w_rd1 = lds0 + W_OFF + (unsigned)wc * 16384u + lane_h1;

  f32x4v acc[8][8];
  bf16x8 fa[2][8], fw[2][8];   // [k-half][i | j]

  auto read_frag = [&](auto h_c, auto t_c) {   // W fragments first: the first MFMA row of a phase needs all of them
    constexpr int Hh = decltype(h_c)::value, t = decltype(t_c)::value;
    if constexpr (t < 8) g4_read<t * 2048>(fw[Hh][t], Hh ? w_rd1 : w_rd0);
    else g4_read<(t - 8) * 2048>(fa[Hh][t - 8], Hh ? a_rd1 : a_rd0);
  };
  auto dma_piece = [&](auto t_c, const char* ab, const char* wb) {
    constexpr int t = decltype(t_c)::value;
    if constexpr (t < 8) g4_dma(voa[t], ab);
    else g4_dma(vow[t - 8], wb);
  };

  // One K tile = 128 MFMA slots (0..63 on k-half 0, 64..127 on k-half 1) and three barriers, so that a buffer's halves are
  // refilled as early as their last reader allows and a request has more than a whole K tile to land:
  //   slots   1..15   8 ds_read: k-half 1 of W (tile t)            -> lgkmcnt(0), barrier 1: the W half of tile t's buffer is free
  //   slots  20..76   8 LDS-DMA, one per 8 MFMAs: W rows of tile t + 2 into it; slots 21..42: 8 ds_read: k-half 1 of A
  //                                                                              -> barrier 2 (slot 47): the A half is free
  //   slots  48..104  8 LDS-DMA, one per 8 MFMAs: A rows of tile t + 2   (two groups of eight pieces right behind the barriers,
  //                   as the vendor kernel issues them, cost 8-9 % at the qkv / w13 shapes: profiles/r04_gemm_ab.md)
  //   slot   87       s_waitcnt vmcnt(13) [everything older than the 13 pieces of tile t + 2 requested so far: my share of tile
  //                   t + 1]; barrier 3
  //   slots  89..119  16 ds_read: k-half 0 of tile t + 1 (other buffer)           -> lgkmcnt(0)
  constexpr int B1 = 19, B2 = 47, B3 = 87;
  auto k_tile = [&](unsigned dst, const char* ab, const char* wb) {
    g4_static_for<0, 128>([&](auto g_c) {
      constexpr int g = decltype(g_c)::value;
      constexpr int Hh = g >> 6, i = (g & 63) / 8, j = g % 8;
      constexpr int rd_w1 = (g >= 1 && g <= 15 && (g & 1)) ? (g - 1) / 2 : -1;                       // W k-half 1, fragment j
      constexpr int dma_w = (g >= 20 && g <= 76 && (g - 20) % 8 == 0) ? (g - 20) / 8 : -1;           // W piece 0..7
      constexpr int rd_a1 = (g >= 21 && g <= 42 && (g - 21) % 3 == 0) ? (g - 21) / 3 : -1;           // A k-half 1, fragment i
      constexpr int dma_a = (g >= 48 && g <= 104 && (g - 48) % 8 == 0) ? (g - 48) / 8 : -1;          // A piece 0..7 (slots 48, 56, ..., 104: none at B3 = 87)
      constexpr int rd_0 = (g >= 89 && g <= 119 && ((g - 89) & 1) == 0) ? (g - 89) / 2 : -1;         // k-half 0 of the next tile
      if constexpr (dma_w >= 0) g4_set_m0(dst + W_OFF + (unsigned)dma_w * 1024u);
      if constexpr (dma_a >= 0) g4_set_m0(dst + (unsigned)dma_a * 1024u);
      g4_mfma(acc[i][j], fw[Hh][j], fa[Hh][i]);
      if constexpr (dma_w >= 0) dma_piece(std::integral_constant<int, 8 + (dma_w < 0 ? 0 : dma_w)>{}, ab, wb);
      if constexpr (dma_a >= 0) dma_piece(std::integral_constant<int, (dma_a < 0 ? 0 : dma_a)>{}, ab, wb);
      if constexpr (rd_w1 >= 0) read_frag(std::integral_constant<int, 1>{}, std::integral_constant<int, (rd_w1 < 0 ? 0 : rd_w1)>{});
      if constexpr (rd_a1 >= 0) read_frag(std::integral_constant<int, 1>{}, std::integral_constant<int, 8 + (rd_a1 < 0 ? 0 : rd_a1)>{});
      if constexpr (rd_0 >= 0) read_frag(std::integral_constant<int, 0>{}, std::integral_constant<int, (rd_0 < 0 ? 0 : rd_0)>{});
      if constexpr (g == B1 || g == B2) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
      // at B3 the 16 pieces of tile t + 2 requested so far in this K tile number 13 (8 of W, 5 of A; the last 3 A pieces follow)
      if constexpr (g == B3) asm volatile("s_waitcnt vmcnt(13)\n\ts_barrier" ::: "memory");
      if constexpr (g == B2) { a_rd1 ^= BUF; w_rd1 ^= BUF; }
    });
    a_rd0 ^= BUF;
    w_rd0 ^= BUF;
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  };

  // ---- prologue: K tiles 0 and 1 of the stream, fragments of k-half 0 of tile 0 ----
  int vid = st_vid;
  stage_setup();
  unsigned dq = 0;   // K tiles requested so far; tile s lives in buffer s & 1
  auto dma_tile_now = [&]() {
    const char* ab = st_a + stage_k_bytes();
    const char* wb = st_w + stage_k_bytes();
    const unsigned dst = lds_wave + (dq & 1u) * BUF;
    g4_static_for<0, 16>([&](auto t_c) {
      constexpr int t = decltype(t_c)::value;
      g4_set_m0(dst + (t < 8 ? 0u : W_OFF) + (unsigned)(t & 7) * 1024u);
      asm volatile("s_nop 0");
      dma_piece(t_c, ab, wb);
    });
    ++dq;
  };
  dma_tile_now();
  stage_next_k_tile();
  dma_tile_now();
  stage_next_k_tile();           // the cursor now stands on stream position 2: what phase B of position 0 requests
  asm volatile("s_waitcnt vmcnt(16)\n\ts_barrier" ::: "memory");
  g4_static_for<0, 16>([&](auto t_c) { read_frag(std::integral_constant<int, 0>{}, t_c); });
  a_rd0 ^= BUF;
  w_rd0 ^= BUF;
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");

  for (;;) {
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
      for (int j = 0; j < 8; ++j) acc[i][j] = f32x4v{0.f, 0.f, 0.f, 0.f};
    asm volatile("s_nop 7" ::: "memory");
    // a wait hipcc can SEE, once per tile, for the previous epilogue's stores - otherwise it parks a vmcnt(0) in front of the first
    // fragment read inside the K loop, where it would also wait for the LDS-DMA it cannot see
    __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0)
    for (int kt = 0; kt < nk; ++kt) {
      k_tile(lds_wave + (dq & 1u) * BUF, st_a + stage_k_bytes(), st_w + stage_k_bytes());
      ++dq;
      stage_next_k_tile();
    }
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");   // the last MFMAs retire before the compiler's accumulator reads
    {
      int tm, tn;
      gemm_tile_coords(p, vid, tm, tn);
      const int64_t mw = (int64_t)tm * 256 + wr * 128, nw = (int64_t)tn * 256 + wc * 128;
      g4k_epilogue<EPI>(p, acc, mw, nw, r16, q);
    }
    vid += stride;
    if (vid >= vid_end) break;
    // the next tile's first fragments were read by the last phase B; reading them AGAIN here makes those registers dead across the
    // epilogue (otherwise hipcc keeps them live through it and spills the epilogue's own values)
    a_rd0 ^= BUF;
    w_rd0 ^= BUF;
    g4_static_for<0, 16>([&](auto t_c) { read_frag(std::integral_constant<int, 0>{}, t_c); });
    a_rd0 ^= BUF;
    w_rd0 ^= BUF;
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the surplus DMA pieces land before the LDS allocation is released
}
