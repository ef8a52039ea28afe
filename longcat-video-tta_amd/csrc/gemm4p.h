// 256 x 256 x 64 tile, 8 waves, TWO phases per K tile: the 8-phase kernel (gemm.hip) with its MFMA clusters doubled.
// Included by gemm.hip (uses GemmParams, gemm_tile_coords, gemm16_epilogue, wait_vmcnt, the LDS image of gemm8p_nt_kernel).
//
// Why: in the ping-pong schedule the two wave rows of a SIMD alternate MFMA clusters, and every cluster pays a fixed gap
// (barrier release, lgkmcnt, issue ramp: ~150 cycles) - with 16-MFMA clusters (256 cycles) the matrix pipe is 0.635 busy (PMC,
// profiles/r02_gemm4w_lab.md), the vendor kernel 0.80.  Here a cluster is 32 MFMAs = one 64 x 64 half of the wave's 128 x 64
// output against BOTH 32-column halves of W, so there are half as many barriers per MFMA; registers as before (one 64-row set
// of A fragments, both W sets).
//     phase A(kt): reads W-nq0, W-nq1, A-mq0 of tile kt (16 ds_read_b128) | wait | B1 | stage A-mq1(kt+1)           | 32 MFMAs | B2
//     phase B(kt): reads A-mq1 of tile kt (8)                             | wait | B1 | stage A-mq0, W-nq0, W-nq1(kt+2) | 32 MFMAs | B2
// The lower wave row (waves 4-7) runs ONE BARRIER behind the upper one, as in the 8-phase kernel.
// Hazards (slots as there: 2 K-tile buffers x {A-mq0, A-mq1, W-nq0, W-nq1}):
//   WAR: a slot is restaged in the phase AFTER the one that read it, and the DMA is issued after that phase's first barrier: the
//        other wave row is then past the second barrier of the reading phase, i.e. past the MFMAs that consumed the reads.
//   RAW: the wait of a phase (before its first barrier) retires everything the NEXT phase reads: in A(kt) all but the 6 pieces
//        of B(kt-1) (so A-mq1(kt) is in), in B(kt) all but the 2 pieces of A(kt) (so tile kt+1's three slots, staged in
//        B(kt-1), are in); a reader passes at least one more barrier than any waiter.  Every slot has two phases to land.
#pragma once

template <int EPI, bool PERSIST>
__global__ __launch_bounds__(512) void gemm4p_nt_kernel(const GemmParams p) {
  constexpr int BUF_BYTES = 65536, SLOT_BYTES = 16384;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 2, wc = wave & 3;
  const int r16 = lane & 15, q = lane >> 4;
  const int nwg = p.vid_begin + p.vid_count;
  const int nk = p.nk1 + p.nk2;   // >= 2, nk1 >= 2 (host-checked)

  // ---- LDS-DMA roles and the LDS image: exactly gemm8p_nt_kernel's ----
  int arow[2][2], wrow[2][2];  // [mq | nq][t]
  unsigned swz[2];
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    const int srow = 8 * (2 * wave + t) + (lane >> 3);
    swz[t] = (unsigned)(((lane & 7) ^ ((srow >> 1) & 7)) * 16);
  }
  int64_t m0 = 0, n0 = 0;
  auto setup_tile = [&](int vid) {
    int tm, tn;
    gemm_tile_coords(p, vid, tm, tn);
    m0 = (int64_t)tm * 256;
    n0 = (int64_t)tn * 256;
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      const int srow = 8 * (2 * wave + t) + (lane >> 3);
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        int64_t g = m0 + (srow >> 6) * 128 + h * 64 + (srow & 63);
        arow[h][t] = (int)(g > p.M - 1 ? p.M - 1 : g);
        g = n0 + (srow >> 5) * 64 + h * 32 + (srow & 31);
        wrow[h][t] = (int)(g > p.N - 1 ? p.N - 1 : g);
      }
    }
  };
  int vid = p.vid_begin + (int)blockIdx.x;
  setup_tile(vid);
  auto stage_a = [&](auto mq_c, int kts, int buf) {
    constexpr int mq = decltype(mq_c)::value;
    const bool lora = kts >= p.nk1;  // the rank-r pair (a2, w2) supplies the last nk2 K tiles
    const char* base = lora ? (const char*)p.a2 + (int64_t)(kts - p.nk1) * 128 : (const char*)p.a + (int64_t)kts * 128;
    const unsigned ldb = (unsigned)(lora ? p.lda2 : p.lda) * 2u;
    unsigned char* dst = smem + buf * BUF_BYTES + mq * SLOT_BYTES + wave * 2048;
#pragma unroll
    for (int t = 0; t < 2; ++t)
      __builtin_amdgcn_global_load_lds((gbl_void*)(base + ((unsigned)arow[mq][t] * ldb + swz[t])), (lds_void*)(dst + t * 1024),
                                       16, 0, 0);
  };
  auto stage_w = [&](auto nq_c, int kts, int buf) {
    constexpr int nq = decltype(nq_c)::value;
    const bool lora = kts >= p.nk1;
    const char* base = lora ? (const char*)p.w2 + (int64_t)(kts - p.nk1) * 128 : (const char*)p.w + (int64_t)kts * 128;
    const unsigned ldb = (unsigned)(lora ? p.ldw2 : p.ldw) * 2u;
    unsigned char* dst = smem + buf * BUF_BYTES + (2 + nq) * SLOT_BYTES + wave * 2048;
#pragma unroll
    for (int t = 0; t < 2; ++t)
      __builtin_amdgcn_global_load_lds((gbl_void*)(base + ((unsigned)wrow[nq][t] * ldb + swz[t])), (lds_void*)(dst + t * 1024),
                                       16, 0, 0);
  };

  const int sw = (r16 >> 1) & 7;
  int a_rd[2], w_rd[2];  // [ks]
#pragma unroll
  for (int ks = 0; ks < 2; ++ks) {
    a_rd[ks] = (wr * 64 + r16) * 128 + ((4 * ks + q) ^ sw) * 16;
    w_rd[ks] = 2 * SLOT_BYTES + (wc * 32 + r16) * 128 + ((4 * ks + q) ^ sw) * 16;
  }

  f32x4v acc[8][4];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4v{0.f, 0.f, 0.f, 0.f};
  bf16x8 af[4][2];     // [i][ks]      activations of the current 64-row half (B operand)
  bf16x8 wf[2][2][2];  // [nq][j][ks]  weights of both 32-column halves (A operand)

  using C0 = std::integral_constant<int, 0>;
  using C1 = std::integral_constant<int, 1>;
  // HALF = 0: phase A (mq = 0, reads both W halves), 1: phase B (mq = 1).  STAGE: issue this phase's pieces; VM: vmcnt to wait for
  auto phase = [&](auto HALF_c, auto STAGE_c, auto VM_c, int kt, int buf) {
    constexpr int HALF = decltype(HALF_c)::value, VM = decltype(VM_c)::value;
    constexpr bool STAGE = decltype(STAGE_c)::value != 0;
    if constexpr (HALF == 0) {
#pragma unroll
      for (int nq = 0; nq < 2; ++nq)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
          for (int ks = 0; ks < 2; ++ks)
            wf[nq][j][ks] = *reinterpret_cast<const bf16x8*>(smem + w_rd[ks] + nq * SLOT_BYTES + j * 2048);
      __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int ks = 0; ks < 2; ++ks)
        af[i][ks] = *reinterpret_cast<const bf16x8*>(smem + a_rd[ks] + HALF * SLOT_BYTES + i * 2048);
    if constexpr (VM >= 0) wait_vmcnt<VM>();
    __builtin_amdgcn_s_barrier();
    if constexpr (STAGE) {   // after the barrier: the other wave row is past the MFMAs that read what is overwritten here
      if constexpr (HALF == 0) {
        stage_a(C1{}, kt + 1, buf ^ 1);
      } else {
        stage_a(C0{}, kt + 2, buf);
        stage_w(C0{}, kt + 2, buf);
        stage_w(C1{}, kt + 2, buf);
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int nq = HALF == 0 ? h : 1 - h;   // A: nq 0, 1;  B: nq 1, 0 (the 8-phase quadrant order)
#pragma unroll
      for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j)
            acc[4 * HALF + i][2 * nq + j] =
                __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[nq][j][ks], af[i][ks], acc[4 * HALF + i][2 * nq + j], 0, 0, 0);
    }
    __builtin_amdgcn_s_setprio(0);
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
  };
  using V8 = std::integral_constant<int, 8>;
  using V6 = std::integral_constant<int, 6>;
  using V2 = std::integral_constant<int, 2>;
  using V0 = std::integral_constant<int, 0>;
  using VN = std::integral_constant<int, -1>;
  auto flip = [&]() {
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) { a_rd[ks] ^= BUF_BYTES; w_rd[ks] ^= BUF_BYTES; }
  };

  // ---- prologue in the steady-state issue order: [A-mq0, W-nq0, W-nq1](0), A-mq1(0), [A-mq0, W-nq0, W-nq1](1) ----
  auto prologue = [&]() {
    stage_a(C0{}, 0, 0);
    stage_w(C0{}, 0, 0);
    stage_w(C1{}, 0, 0);
    stage_a(C1{}, 0, 0);
    stage_a(C0{}, 1, 1);
    stage_w(C0{}, 1, 1);
    stage_w(C1{}, 1, 1);
  };
  prologue();
  for (;;) {
    // the three slots phase A(0) reads have landed (8 younger pieces may be in flight; persistent: the previous tile's stores
    // are younger still, so this is conservative, never early)
    wait_vmcnt<8>();
    __builtin_amdgcn_s_barrier();
    if (wr == 1) __builtin_amdgcn_s_barrier();  // the lower wave row runs one barrier behind

    int kt = 0;
    for (; kt < nk - 2; ++kt) {
      const int buf = kt & 1;
      phase(C0{}, C1{}, V6{}, kt, buf);
      phase(C1{}, C1{}, V2{}, kt, buf);
      flip();
    }
    {  // K tile nk-2: A-mq1 of the last tile is still to stage; there is no tile nk
      const int buf = kt & 1;
      phase(C0{}, C1{}, V6{}, kt, buf);
      phase(C1{}, C0{}, V2{}, kt, buf);
      flip();
      ++kt;
    }
    {  // K tile nk-1
      const int buf = kt & 1;
      phase(C0{}, C0{}, V0{}, kt, buf);
      phase(C1{}, C0{}, VN{}, kt, buf);
    }
    if (wr == 0) __builtin_amdgcn_s_barrier();  // balance the stagger

    const int64_t mw = m0 + wr * 128, nw = n0 + wc * 64;
    if constexpr (PERSIST) {
      const int next = vid + (int)gridDim.x;
      const bool more = next < nwg;
      if (more) {
        if (kt & 1) flip();  // fragment read bases back to buffer 0
        setup_tile(next);
        prologue();
      }
      gemm16_epilogue<8, 4, EPI>(p, acc, mw, nw, r16, q);
      if (!more) break;
      vid = next;
#pragma unroll
      for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4v{0.f, 0.f, 0.f, 0.f};
    } else {
      gemm16_epilogue<8, 4, EPI>(p, acc, mw, nw, r16, q);
      break;
    }
  }
}
