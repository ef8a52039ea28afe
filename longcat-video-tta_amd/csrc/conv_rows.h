// Row-tile convolution for the VAE's 96-channel stages (included by gemm.hip: GemmParams, gemm16_epilogue, f32x4v, bf16x8).
//
// Why a second convolution kernel.  The implicit-GEMM form above stages, for every tap, the 128 or 256 gathered input rows of
// its output tile: at 96 channels (the 720p stage of the decoder, the first stage of the encoder) that is a 128 x 128 tile
// padded from 96 x 96 - 44 % of the MFMA work multiplies zeros - and 32 KB of L2->LDS traffic per 1.2 real MFLOP.  Counters
// (profiles/r02_conv_rows.md): MFMA busy 0.32, HBM 0.7 TB/s, L2 hit 0.92, 11.4 TB/s of LDS-DMA fill = ~20 B/clk per CU, the rate
// every LDS-staged kernel of this library tops out at - the kernel is bound by the L2->LDS fill path, not by the matrix pipe,
// HBM or address arithmetic.
//
// This kernel cuts that traffic 3.6x and drops the padding:
//   * an output tile is 256 consecutive pixels of ONE image row (b, t, h, w0 .. w0+255), all Cout <= 96 channels;
//   * a step is one (dt, dh, 96-channel slice): the input row segment [w0 - pw, w0 + 256 + kw - 1 - pw) is staged ONCE
//     (258 pixels x 96 channels) and serves the kw taps of that row - tap dw of output pixel r reads staged pixel r + dw;
//     row / frame validity of a step is uniform over the tile (skipped steps are never staged), column validity is the range
//     check of the buffer load that stages a lane's 16 bytes (outside: zeros), so the K loop carries no per-pixel masks at all;
//   * the weights of one tap (Cout x 96) are staged per tap, in three buffers beside the double-buffered pixel segment;
//   * with the folded 2x upsample the segment is staged in UPSAMPLED coordinates (pixel u reads source u >> 1).
//
// LDS image.  The fragment of lane (r16, q) for K block kk is 16 bytes; a tap reads the same image at a row offset dw, so the
// image must be conflict-free for ds_read_b128 under ANY row shift.  An XOR swizzle is not (it is tied to row mod 16), and a
// padded row stride cannot be (the four 16-lane groups of ds_read_b128 mix rows {0-3, 12-15} of q with rows {4-11} of q + 1).
// What is: lanes q and q + 1 (never q and q + 2) meet in a lane group, so one plane per PARITY of q, rows 96 bytes apart, the
// second plane 16 (mod 32) bytes behind a multiple of 256 -
//     addr(row, q, kk) = (q & 1) * PLANE + row * 96 + (q >> 1) * 48 + kk * 16
// Bank unit (addr / 16) mod 16 = (6 row + 3 (q >> 1) + kk + (q & 1)) mod 16: 6 row takes the 8 even values over the 8 rows a
// lane group reads from one plane (rows {0-3, 12-15} or {4-11}, shifted by anything), and the other plane's 8 rows land on
// the odd ones.  The 96 bytes of a plane row are channels [48 (q & 1), +48): the MFMA K index is permuted (k = 8 (l >> 4) + e
// of block kk  <->  channel 48 (q & 1) + 24 (q >> 1) + 8 kk + e) identically for both operands, which changes only the order
// of the fp32 sums.  LDS-DMA fills a plane linearly: lane l of piece c writes unit 64 c + l = 16-byte block (64 c + l) % 6 of
// row (64 c + l) / 6, so one request covers ~11 pixel rows with 96 contiguous bytes each (a first version with four planes of
// 48-byte rows touched 22-43 cache lines per request).
#pragma once
#ifndef CONV_ROWS_LOADERS
#define CONV_ROWS_LOADERS 8
#endif

template <int WM, int WN, int TM, int TN>
struct ConvRowsCfg {
  static constexpr int BM = WM * TM * 16;           // 256 output pixels
  static constexpr int BN = WN * TN * 16;           // 96 (or 16: the 3-channel head)
  static constexpr int NW = WM * WN;                // 8 waves
  static constexpr int A_PIECES_PER_PLANE = 25;     // 25 KiB >= (256 + 2) rows x 96 B
  static constexpr int PLA = A_PIECES_PER_PLANE * 1024 + 16;   // plane offset = 16 (mod 32) bytes: see the bank argument
  static constexpr int A_BYTES = 2 * PLA;           // 51232
  static constexpr int A_PIECES = 2 * A_PIECES_PER_PLANE;
  static constexpr int B_PIECES_PER_PLANE = (BN * 96 + 1023) / 1024;   // 9 / 2
  static constexpr int PLB = B_PIECES_PER_PLANE * 1024 + 16;
  static constexpr int B_BYTES = 2 * PLB;
  static constexpr int B_PIECES = 2 * B_PIECES_PER_PLANE;   // 18 / 4
  static constexpr int IMAGE_BYTES = 2 * A_BYTES + 3 * B_BYTES;   // 157856
  static constexpr int MAX_STEPS = 32, MAX_SUB = 64;             // table sizes (host-checked against kt * kh * slices, x kw)
  static constexpr int TABLE = (IMAGE_BYTES + 15) / 16 * 16;
  static constexpr int TABLE_BYTES = (MAX_STEPS + MAX_SUB) * 8;   // per tile parity
  static constexpr int LDS_BYTES = TABLE + 2 * TABLE_BYTES;       // 159392 of 163840
  static_assert(BM == 256 && NW == 8, "conv_rows: 256-pixel tiles, 8 waves");
};

struct ConvRowsTile {          // what the sub-step loop and the epilogue need of one output tile (wave-uniform)
  int nsteps, nsub;            // steps = valid (dt, dh) x channel slices; sub-steps = steps x kw
  int w0;                      // first output column
  int64_t row_begin;           // output row index m of column 0 of this image row
};

// Persistent: gridDim.x = 8 G workgroups, one per CU.  Workgroup (xcd = id % 8, slot = id / 8) walks tiles slot, slot + G, ...
// of its XCD's contiguous run of the tile sequence, so the G workgroups of an XCD always work on G neighbouring tiles (a band
// of image rows whose dh / dt neighbours stay in that XCD's L2) and the sub-step stream never stops: the first segment and
// the first two taps' weights of the NEXT tile are requested during the last sub-steps of the current one and land while its
// epilogue runs.
//
// Producer / consumer waves.  The CU's LDS-DMA path fills at ~20-25 B/clk (the rate every LDS-staged kernel of this library
// tops out at), i.e. ~1600 cycles for the 35 KB of a sub-step, and a wave that issues a request while the path is busy is HELD
// at that instruction: stamps of an earlier form, where the eight MFMA waves issued their own requests, showed 900-1300 cycles
// per wave and sub-step inside the 5-7 request instructions, ~1150 in reads + MFMAs, and a sub-step of ~3300 cycles - request
// and multiply time simply added up, whatever their order.  Now waves 8-11 (one per SIMD) do nothing but requests - they keep
// the fill path busy and absorb its back-pressure - and waves 0-7 do nothing but fragment reads and MFMAs; one barrier per
// sub-step hands buffers over (the loaders' counted vmcnt before it says "landed", the MFMA waves' arrival says "consumed").
//
// The sub-step loop is kept free of index arithmetic: loader 0 writes, once per tile, the source base of every step (pixel
// segment) and sub-step (weights) into a small LDS table - one lane per sub-step, the divisions run 64 wide - and the loaders
// advance cursors over the tables of the current and the next tile.
template <int WM, int WN, int TM, int TN, int EPI>
__global__ __launch_bounds__((8 + CONV_ROWS_LOADERS) * 64) void conv_rows_kernel(const GemmParams p, const int tiles_per_row, const int n_tiles) {
  using Cfg = ConvRowsCfg<WM, WN, TM, TN>;
  constexpr int NL = CONV_ROWS_LOADERS;                            // loader waves
  constexpr int LA = (Cfg::A_PIECES + NL - 1) / NL;                // 13 piece slots of the pixel segment per loader
  constexpr int LB = (Cfg::B_PIECES + NL - 1) / NL;                // 5 / 1 of a tap's weights
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const bool loader = wave >= Cfg::NW;
  const int lw = wave - Cfg::NW;                                   // loader index (loaders only)
  const int kw = p.cv_kw;

  // this workgroup's tiles: first + k * stride while < last
  int tile_next, tile_end;
  const int tile_stride = (int)gridDim.x >> 3;
  {
    const int xcd = (int)blockIdx.x & 7, slot = (int)blockIdx.x >> 3;
    const int per = n_tiles >> 3, r8 = n_tiles & 7;
    const int begin = xcd < r8 ? xcd * (per + 1) : r8 * (per + 1) + (xcd - r8) * per;
    tile_end = begin + per + (xcd < r8 ? 1 : 0);
    tile_next = begin + slot;
  }
  if (tile_next >= tile_end) return;                             // (whole workgroup)

  // ---- per-tile plan ----
  // Every wave: the scalars (steps, sub-steps, where the tile's rows are).  Loaders: a_off = where each lane's LDS-DMA pieces
  // of the pixel segment come from, relative to the start of the input image row (2^31: outside - the buffer range check
  // returns zeros); piece slot t of loader l is piece g = l + 4 t = unit (g % 25) * 64 + lane of plane g / 25 = 16-byte block
  // u % 6 of staged row u / 6.  Loader 0: the tables (LDS, per tile parity): tab_a[step] = address of channel slice cs of input
  // image row (t0 + dt, (h0 + dh) >> up), tab_b[sub-step] = address of w[0][tap][96 cs]; steps enumerate the (dt, dh) taps
  // that read inside the tensor (ranges: padding is in front in t, symmetric in h, the stride is 1) times the channel slices.
  auto tab_a = [&](int parity, int i) { return reinterpret_cast<uint64_t*>(smem + Cfg::TABLE + parity * Cfg::TABLE_BYTES) + i; };
  auto tab_b = [&](int parity, int i) { return tab_a(parity, Cfg::MAX_STEPS + i); };
  auto plan_tile = [&](int pid, int parity, ConvRowsTile& c, unsigned (&a_off)[LA]) {
    const int up = p.cv_up2x ? 1 : 0;
    const int ldx = (int)p.lda, ncs = p.cv_cpt;
    // tile sequence: frame index innermost.  The G workgroups of an XCD then work on G consecutive frames of the same row
    // segment at the same time, and the three dt taps of neighbouring tiles read the same input rows out of that XCD's L2
    // (with w innermost every input row was fetched from HBM once per dt tap: fetch = 2.6 x the input)
    int to, wt, ho, bb;
    if (p.splitk == 0) {
      to = pid % p.cv_T;
      int r = pid / p.cv_T;
      wt = r % tiles_per_row; r /= tiles_per_row;
      ho = r % p.cv_H;
      bb = r / p.cv_H;
    } else {                                                     // LCV_CONV_ROWS_ORDER=w (A/B): w innermost, then h, t
      wt = pid % tiles_per_row;
      int r = pid / tiles_per_row;
      ho = r % p.cv_H; r /= p.cv_H;
      to = r % p.cv_T;
      bb = r / p.cv_T;
    }
    c.row_begin = ((int64_t)(bb * p.cv_T + to) * p.cv_H + ho) * p.cv_W;
    c.w0 = wt * Cfg::BM;
    const int t0 = to - p.cv_pt, h0 = ho - p.cv_ph;
    const int hb = p.cv_Hin << up;
    const int dt_lo = t0 < 0 ? -t0 : 0;
    const int dt_hi = (p.cv_Tin - t0) < p.cv_kt ? (p.cv_Tin - t0) : p.cv_kt;
    const int dh_lo = h0 < 0 ? -h0 : 0;
    const int dh_hi = (hb - h0) < p.cv_kh ? (hb - h0) : p.cv_kh;
    const int ndh = dh_hi - dh_lo;
    c.nsteps = (dt_hi - dt_lo) * ndh * ncs;
    c.nsub = c.nsteps * kw;
    if (!loader) return;
    const int wbound = p.cv_Win << up;
#pragma unroll
    for (int t = 0; t < LA; ++t) {
      const int g = t * NL + lw;
      const int plane = g / Cfg::A_PIECES_PER_PLANE;
      const int u = (g - plane * Cfg::A_PIECES_PER_PLANE) * 64 + lane;
      const int row = u / 6, blk = u - row * 6;
      const int uu = c.w0 - p.cv_pw + row;                       // (upsampled) input column of staged row `row`
      const bool ok = g < Cfg::A_PIECES && row < Cfg::BM + kw - 1 && uu >= 0 && uu < wbound;
      a_off[t] = ok ? (unsigned)(((uu >> up) * ldx + plane * 48 + blk * 8) * 2) : 0x80000000u;   // outside: past any num_records
    }
    if (lw == 0) {                                               // lane l: sub-step l
      if (lane < c.nsub) {
        const int step = lane / kw, dw = lane - step * kw;
        const int cs = step % ncs, r = step / ncs;
        const int dh = dh_lo + r % ndh, dt = dt_lo + r / ndh;
        const int tap = (dt * p.cv_kh + dh) * kw + dw;
        *tab_b(parity, lane) = (uint64_t)(p.w + (int64_t)tap * ldx + cs * 96);
        if (dw == 0) {
          const int64_t in_row = ((int64_t)(bb * p.cv_Tin + t0 + dt) * p.cv_Hin + ((h0 + dh) >> up)) * p.cv_Win;
          *tab_a(parity, step) = (uint64_t)(p.a + in_row * ldx + cs * 96);
        }
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");         // written before this wave's next barrier
    }
  };

  if (loader) {
    // =========================== loader waves ===========================
    auto table_entry = [&](const uint64_t* e) {                  // uniform LDS read -> scalar registers
      const uint64_t v = *e;
      const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
      return ((uint64_t)hi << 32) | lo;
    };
    // weights of one tap: piece g = l + 4 t, the same unit order over Cout rows (tile-independent)
    unsigned b_off[LB];
#pragma unroll
    for (int t = 0; t < LB; ++t) {
      const int g = t * NL + lw;
      const int plane = g / Cfg::B_PIECES_PER_PLANE;
      const int u = (g - plane * Cfg::B_PIECES_PER_PLANE) * 64 + lane;
      int n = u / 6;
      const int blk = u - n * 6;
      if (n > (int)p.N - 1) n = (int)p.N - 1;                    // rows past Cout: any finite data, masked by the epilogue
      b_off[t] = (unsigned)((n * (int)p.ldw + plane * 48 + blk * 8) * 2);
    }
    // Requests are buffer loads into LDS (buffer_load_dwordx4 ... offen lds): a 128-bit resource built from the scalar table
    // entry, the lane's 32-bit offset from the plan, and the hardware range check as the zero padding - a lane whose offset
    // is past num_records (the "outside" mark) gets zeros written to its LDS slot.  No vector instruction per piece.
    const int row_bytes = p.cv_Win * (int)p.lda * 2;             // one input image row
    const int w_bytes = (int)(p.N * p.ldw * 2);
    auto stage_pixels = [&](uint64_t base, const unsigned (&a_off)[LA], int buf, int t_begin, int t_end) {
      const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, row_bytes, 0x00020000);
      unsigned char* dst = smem + buf * Cfg::A_BYTES;
#pragma unroll
      for (int t = 0; t < LA; ++t) {
        if (t < t_begin || t >= t_end) continue;
        const int g = t * NL + lw;
        if (g >= Cfg::A_PIECES) continue;                        // wave-uniform
        const int plane = g / Cfg::A_PIECES_PER_PLANE;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_void*)(dst + plane * Cfg::PLA + (g - plane * Cfg::A_PIECES_PER_PLANE) * 1024), 16,
                                                 (int)a_off[t], 0, 0, 0);
      }
    };
    auto stage_weights = [&](uint64_t base, int buf) {
      const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, w_bytes, 0x00020000);
      unsigned char* dst = smem + 2 * Cfg::A_BYTES + buf * Cfg::B_BYTES;
#pragma unroll
      for (int t = 0; t < LB; ++t) {
        const int g = t * NL + lw;
        if (g >= Cfg::B_PIECES) continue;                        // wave-uniform
        const int plane = g / Cfg::B_PIECES_PER_PLANE;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_void*)(dst + plane * Cfg::PLB + (g - plane * Cfg::B_PIECES_PER_PLANE) * 1024), 16,
                                                 (int)b_off[t], 0, 0, 0);
      }
    };
    // how many requests this loader makes: per tap of weights, and per part of a segment (the segment of the NEXT step is
    // requested in parts over the sub-steps of the current one: 5 / 4 / 4 piece slots with three taps along w)
    auto count = [&](int t_begin, int t_end, int pieces) {
      int n = 0;
      for (int t = t_begin; t < t_end; ++t) n += (t * NL + lw < pieces) ? 1 : 0;
      return n;
    };
    constexpr int P0 = (LA * 5 + 12) / 13, P1 = P0 + (LA - P0 + 1) / 2, PH = (LA + 1) / 2;   // 5 | 9 of 13 slots; 7 of 13
    auto part_begin = [&](int dw) { return kw >= 3 ? (dw == 0 ? 0 : dw == 1 ? P0 : dw == 2 ? P1 : LA) : kw == 2 ? (dw == 0 ? 0 : PH) : (dw == 0 ? 0 : LA); };
    auto part_end = [&](int dw) { return kw >= 3 ? (dw == 0 ? P0 : dw == 1 ? P1 : LA) : kw == 2 ? (dw == 0 ? PH : LA) : LA; };
    const int nb = count(0, LB, Cfg::B_PIECES);
    const int na0 = count(part_begin(0), part_end(0), Cfg::A_PIECES), na1 = count(part_begin(1), part_end(1), Cfg::A_PIECES),
              na2 = count(part_begin(2), part_end(2), Cfg::A_PIECES);
    auto wait_allow = [&](int n) {     // s_waitcnt takes an immediate; at most LA + LB = 18 requests per sub-step
      switch (n) {
        case 0: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
        case 1: asm volatile("s_waitcnt vmcnt(1)" ::: "memory"); break;
        case 2: asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); break;
        case 3: asm volatile("s_waitcnt vmcnt(3)" ::: "memory"); break;
        case 4: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
        case 5: asm volatile("s_waitcnt vmcnt(5)" ::: "memory"); break;
        case 6: asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); break;
        case 7: asm volatile("s_waitcnt vmcnt(7)" ::: "memory"); break;
        case 8: asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); break;
        case 9: asm volatile("s_waitcnt vmcnt(9)" ::: "memory"); break;
        case 10: asm volatile("s_waitcnt vmcnt(10)" ::: "memory"); break;
        case 11: asm volatile("s_waitcnt vmcnt(11)" ::: "memory"); break;
        case 12: asm volatile("s_waitcnt vmcnt(12)" ::: "memory"); break;
        case 13: asm volatile("s_waitcnt vmcnt(13)" ::: "memory"); break;
        case 14: asm volatile("s_waitcnt vmcnt(14)" ::: "memory"); break;
        case 15: asm volatile("s_waitcnt vmcnt(15)" ::: "memory"); break;
        case 16: asm volatile("s_waitcnt vmcnt(16)" ::: "memory"); break;
        case 17: asm volatile("s_waitcnt vmcnt(17)" ::: "memory"); break;
        default: asm volatile("s_waitcnt vmcnt(18)" ::: "memory"); break;
      }
    };

    ConvRowsTile cur, nxt;
    unsigned a_cur[LA], a_nxt[LA];
    int par = 0;                                                 // table parity of cur
    plan_tile(tile_next, 0, cur, a_cur);
    tile_next += tile_stride;
    bool has_nxt = tile_next < tile_end;
    if (has_nxt) plan_tile(tile_next, 1, nxt, a_nxt);
    else nxt = cur;
    __builtin_amdgcn_s_barrier();                                // [start] tables visible to the other loaders
    stage_pixels(table_entry(tab_a(0, 0)), a_cur, 0, 0, LA);
    stage_weights(table_entry(tab_b(0, 0)), 0);
    stage_weights(table_entry(tab_b(0, 1)), 1);                  // every tile has >= 2 sub-steps (host-checked)
    int allow = nb;
    int abuf = 0;                                                // pixel buffer of the step being multiplied
    // request cursors: weights of sub-step (current + 2), segment of step (current + 1); `_nx`: the cursor is in the next tile
    int rb_idx = 2, rb_buf = 2, ra_idx = 1;
    bool rb_nx = false, ra_nx = false;
    auto normalize = [&]() {
      if (!rb_nx && rb_idx >= cur.nsub) { rb_idx -= cur.nsub; rb_nx = true; }
      if (!ra_nx && ra_idx >= cur.nsteps) { ra_idx -= cur.nsteps; ra_nx = true; }
    };
    normalize();
    for (;;) {
      int dw = 0;
      for (int sub = 0; sub < cur.nsub; ++sub) {
        wait_allow(allow);                                       // this sub-step's images have landed (this loader's part)
        __builtin_amdgcn_s_barrier();                            // [sub-step] hand-over: landed / consumed
        int issued = 0, a_issued = 0;
        if (!ra_nx || has_nxt) {                                 // segment of the next step (older than the weights below)
          const int tb = part_begin(dw), te = part_end(dw);
          if (tb < te) {
            const uint64_t base = table_entry(tab_a(par ^ (ra_nx ? 1 : 0), ra_idx));
            if (ra_nx) stage_pixels(base, a_nxt, abuf ^ 1, tb, te);
            else stage_pixels(base, a_cur, abuf ^ 1, tb, te);
            a_issued = dw == 0 ? na0 : dw == 1 ? na1 : dw == 2 ? na2 : 0;
            issued = a_issued;
          }
        }
        if (!rb_nx || has_nxt) {
          stage_weights(table_entry(tab_b(par ^ (rb_nx ? 1 : 0), rb_idx)), rb_buf);
          issued += nb;
        }
        rb_buf = rb_buf == 2 ? 0 : rb_buf + 1;
        ++rb_idx;
        const bool step_ends = dw == kw - 1;
        allow = step_ends ? issued - a_issued : issued;          // next sub-step starts a step: its segment must be complete
        if (step_ends) { dw = 0; abuf ^= 1; ++ra_idx; } else ++dw;
        normalize();
      }
      if (!has_nxt) break;
      cur = nxt;
#pragma unroll
      for (int t = 0; t < LA; ++t) a_cur[t] = a_nxt[t];
      par ^= 1;
      rb_nx = false; ra_nx = false;                              // the cursors were in this tile already
      tile_next += tile_stride;
      has_nxt = tile_next < tile_end;
      if (has_nxt) plan_tile(tile_next, par ^ 1, nxt, a_nxt);
      normalize();
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    return;
  }

  // =========================== MFMA waves ===========================
  const int wm = wave / WN, wn = wave % WN;
  const int r16 = lane & 15, q = lane >> 4;
  f32x4v acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j) acc[i][j] = f32x4v{0.f, 0.f, 0.f, 0.f};
  const int a_frag = (q & 1) * Cfg::PLA + (wm * TM * 16 + r16) * 96 + (q >> 1) * 48;
  const int b_frag = 2 * Cfg::A_BYTES + (q & 1) * Cfg::PLB + (wn * TN * 16 + r16) * 96 + (q >> 1) * 48;
  ConvRowsTile cur;
  unsigned unused_plan[LA];
  plan_tile(tile_next, 0, cur, unused_plan);
  tile_next += tile_stride;
  __builtin_amdgcn_s_barrier();                                  // [start]
  int abuf = 0, bbuf = 0;                                        // buffers of the step / sub-step being multiplied
  for (;;) {
    int dw = 0;
    for (int sub = 0; sub < cur.nsub; ++sub) {
      __builtin_amdgcn_s_barrier();                              // [sub-step] this sub-step's images have landed
      const unsigned char* sa = smem + abuf * Cfg::A_BYTES + a_frag + dw * 96;
      const unsigned char* sb = smem + bbuf * Cfg::B_BYTES + b_frag;
#pragma unroll
      for (int kk = 0; kk < 3; ++kk) {
        bf16x8 af[TM], bfr[TN];
#pragma unroll
        for (int i = 0; i < TM; ++i) af[i] = *reinterpret_cast<const bf16x8*>(sa + i * 16 * 96 + kk * 16);
#pragma unroll
        for (int j = 0; j < TN; ++j) bfr[j] = *reinterpret_cast<const bf16x8*>(sb + j * 16 * 96 + kk * 16);
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[j], af[i], acc[i][j], 0, 0, 0);  // C^T tile
      }
      bbuf = bbuf == 2 ? 0 : bbuf + 1;
      if (dw == kw - 1) { dw = 0; abuf ^= 1; } else ++dw;
    }
    // rows of this tile are output pixels row_begin + w0 + r; those past the end of the image row do not exist
    {
      GemmParams pe = p;
      pe.M = cur.row_begin + p.cv_W;
      const int64_t mw = cur.row_begin + cur.w0 + wm * TM * 16, nw = wn * TN * 16;
      // full tiles: hoisted row pointers and 8-byte loads / stores (the same expressions as the generic form)
      if (g4_fast_epilogue_ok<EPI, TN, TM>(pe, mw, nw)) g4_fast_epilogue<EPI, TN, TM>(pe, acc, mw, nw, r16, q);
      else gemm16_epilogue<TM, TN, EPI>(pe, acc, mw, nw, r16, q);
    }
    if (tile_next >= tile_end) break;
    plan_tile(tile_next, 0, cur, unused_plan);
    tile_next += tile_stride;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j) acc[i][j] = f32x4v{0.f, 0.f, 0.f, 0.f};
  }
}

template <int WM, int WN, int TM, int TN, int EPI>
static int launch_conv_rows(GemmParams& p, hipStream_t s) {
  using Cfg = ConvRowsCfg<WM, WN, TM, TN>;
  const int tiles_per_row = (p.cv_W + Cfg::BM - 1) / Cfg::BM;
  const int64_t n_tiles = (p.M / p.cv_W) * tiles_per_row;
  LCV_CHECK_ARG(n_tiles < (int64_t(1) << 31), "conv3d: %ld row tiles", (long)n_tiles);
  auto kern = conv_rows_kernel<WM, WN, TM, TN, EPI>;
  // (function-local static: initialised once, thread-safe)
  static const bool attr_ok = !(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, Cfg::LDS_BYTES) != hipSuccess);
  if (!attr_ok) {
      lcv_set_error("conv3d: cannot raise dynamic LDS to %d", Cfg::LDS_BYTES);
      return LCV_EDEVICE;
  }
  // one workgroup per CU, a multiple of 8 (XCDs); fewer when there are fewer tiles
  int grid = 256;
  { const char* e = lcv_knob("LCV_CONV_ROWS_GRID"); if (e && atoi(e) >= 8) grid = atoi(e) / 8 * 8; }
  if ((int64_t)grid > (n_tiles + 7) / 8 * 8) grid = (int)((n_tiles + 7) / 8 * 8);
  hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3((8 + CONV_ROWS_LOADERS) * 64), Cfg::LDS_BYTES, s, p, tiles_per_row, (int)n_tiles);   // 8 MFMA waves + the loaders
  LCV_LAUNCH_CHECK("conv_rows");
  return LCV_OK;
}

// Which convolutions take the row-tile kernel: stride 1, at most 3 taps along w, 96-channel slices, Cout <= 96, and rows long
// enough that the last (partial) tile of a row does not dominate.
static bool conv_rows_applies(const GemmParams& p, int64_t cin) {
  if (lcv_knob("LCV_CONV_ROWS") && atoi(lcv_knob("LCV_CONV_ROWS")) == 0) return false;
  if (p.cv_st != 1 || p.cv_sh != 1 || p.cv_sw != 1) return false;
  if (cin % 96 != 0 || p.N > 96 || p.cv_kw > 3) return false;
  const int slices = (int)(cin / 96);
  if (slices * p.cv_kw * (p.cv_kh > 1 ? 2 : 1) < 2) return false;   // the pipeline wants >= 2 sub-steps in every tile
  if (p.cv_kt * p.cv_kh * slices > 32 || p.cv_kt * p.cv_kh * slices * p.cv_kw > 64) return false;   // plan tables
  const int tiles_per_row = (p.cv_W + 255) / 256;
  return p.cv_W * 4 >= tiles_per_row * 256 * 3;     // >= 75 % of the tile rows are real pixels
}
