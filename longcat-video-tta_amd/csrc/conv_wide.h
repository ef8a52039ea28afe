// Implicit-GEMM convolution for Cout = 192 / 384 with producer / consumer waves (included by gemm.hip after gemm16_epilogue).
//
// Same tile, LDS image, K order and MFMA sequence as gemm16_nt_kernel<256, 192, 2, 4, EPI, true> - bit-identical results - but
// the eight MFMA waves no longer issue the LDS-DMA requests of the next K tile themselves: a wave is held at an LDS-DMA
// instruction while the CU's fill path is busy (profiles/r02_conv_rows.md: 150-200 cycles per request, seven per wave and
// K tile), and its fragment reads and MFMAs queue behind that in program order.  Waves 8-15 (two per SIMD) stage every K tile -
// they carry the gather's per-row state (pixel index of tap (0, 0, 0), validity bit per tap, upsample parity) and absorb the
// back-pressure -, waves 0-7 only read fragments and multiply; one barrier per K tile hands a buffer over (the loaders'
// vmcnt(0) before it says "landed", the MFMA waves' arrival says "consumed").
#pragma once

template <int EPI>
__global__ __launch_bounds__(1024) void conv_wide_kernel(const GemmParams p) {
  constexpr int BM = 256, BN = 192, WR = 2, WC = 4, TM = 8, TN = 3, NL = 8;
  constexpr int A_BYTES = BM * 128, B_BYTES = BN * 128, STAGE_BYTES = A_BYTES + B_BYTES;   // 57344
  constexpr int A_PIECES = BM / 8, B_PIECES = BN / 8;                                      // 1 KiB = 8 rows x 128 B each
  constexpr int LA = A_PIECES / NL, LB = B_PIECES / NL;                                    // 4 + 3 requests per loader and K tile
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  int tm, tn;
  gemm_tile_coords(p, tm, tn);
  const int64_t m0 = (int64_t)tm * BM, n0 = (int64_t)tn * BN;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int nk = p.nk1;

  if (wave >= 8) {
    // =========================== loader waves ===========================
    const int lw = wave - 8;
    const int ld_row = lane >> 3, ld_slot = lane & 7;
    // piece slot t of loader l is piece l + 8 t: rows 8 (l + 8 t) .. + 8 of the A (B) image, lane -> row ld_row, 16-byte slot
    // ld_slot, which holds logical chunk ld_slot ^ ((row >> 1) & 7)
    int pix0[LA];
    unsigned okm[LA], par[LA], a_sw[LA];
    const bf16_t* b_ptr[LB];
#pragma unroll
    for (int t = 0; t < LA; ++t) {
      const int row = (lw + NL * t) * 8 + ld_row;
      int64_t g = m0 + row;
      if (g > p.M - 1) g = p.M - 1;
      a_sw[t] = (unsigned)((ld_slot ^ ((row >> 1) & 7)) * 16);
      const int wo = (int)(g % p.cv_W); g /= p.cv_W;
      const int ho = (int)(g % p.cv_H); g /= p.cv_H;
      const int to = (int)(g % p.cv_T);
      const int bb = (int)(g / p.cv_T);
      // front padding only: causal in t, (k/2 | 0) in h, w
      const int t0 = to * p.cv_st - p.cv_pt, h0 = ho * p.cv_sh - p.cv_ph, w0 = wo * p.cv_sw - p.cv_pw;
      const int hb = p.cv_up2x ? 2 * p.cv_Hin : p.cv_Hin, wb = p.cv_up2x ? 2 * p.cv_Win : p.cv_Win;
      unsigned ok = 0;
      int tap = 0;
      for (int dt = 0; dt < p.cv_kt; ++dt)
        for (int dh = 0; dh < p.cv_kh; ++dh)
          for (int dw = 0; dw < p.cv_kw; ++dw, ++tap) {
            const int ti = t0 + dt, hi = h0 + dh, wi = w0 + dw;
            if (ti >= 0 && ti < p.cv_Tin && hi >= 0 && hi < hb && wi >= 0 && wi < wb) ok |= 1u << tap;
          }
      okm[t] = ok;
      // with the upsample: floor((x + d) / 2) = (x >> 1) + (((x & 1) + d) >> 1) for d >= 0 (arithmetic shift, x may be -1)
      const int hq = p.cv_up2x ? (h0 >> 1) : h0, wq = p.cv_up2x ? (w0 >> 1) : w0;
      par[t] = p.cv_up2x ? (unsigned)((h0 & 1) | ((w0 & 1) << 1)) : 0u;
      pix0[t] = ((bb * p.cv_Tin + t0) * p.cv_Hin + hq) * p.cv_Win + wq;
    }
#pragma unroll
    for (int t = 0; t < LB; ++t) {
      const int row = (lw + NL * t) * 8 + ld_row;
      int64_t g = n0 + row;
      if (g > p.N - 1) g = p.N - 1;
      b_ptr[t] = p.w + g * p.ldw + (ld_slot ^ ((row >> 1) & 7)) * 8;
    }
    // K tiles are staged in order, so the tap coordinates advance by scalar increments
    int c_chunk = 0, c_dw = 0, c_dh = 0, c_dt = 0, c_tap = 0;
    const unsigned row_b = (unsigned)p.lda * 2u;
    const uint64_t zbase = (uint64_t)p.cv_zero;
    auto stage = [&](int kt, int buf) {
      unsigned char* sa = smem + buf * STAGE_BYTES;
      unsigned char* sb = sa + A_BYTES;
      const int tap = c_tap, dh = c_dh, dw = c_dw;
      const int dpix_t = c_dt * p.cv_Hin * p.cv_Win;
      const int dpix = dpix_t + dh * p.cv_Win + dw;
      const uint64_t abase = (uint64_t)(p.a + c_chunk * 64);
      if (++c_chunk == p.cv_cpt) {
        c_chunk = 0; ++c_tap;
        if (++c_dw == p.cv_kw) { c_dw = 0; if (++c_dh == p.cv_kh) { c_dh = 0; ++c_dt; } }
      }
#pragma unroll
      for (int t = 0; t < LA; ++t) {
        int pix;
        if (p.cv_up2x) {
          const int ih = (int)((par[t] & 1u) + (unsigned)dh) >> 1, iw = (int)((par[t] >> 1) + (unsigned)dw) >> 1;
          pix = pix0[t] + dpix_t + __mul24(ih, p.cv_Win) + iw;
        } else {
          pix = pix0[t] + dpix;
        }
        const bool ok = (okm[t] >> tap) & 1u;
        const uint64_t src = (ok ? abase : zbase) + (uint64_t)(ok ? (unsigned)pix : 0u) * row_b + a_sw[t];
        __builtin_amdgcn_global_load_lds((gbl_void*)src, (lds_void*)(sa + (lw + NL * t) * 1024), 16, 0, 0);
      }
      const int k0 = kt * 64;
#pragma unroll
      for (int t = 0; t < LB; ++t)
        __builtin_amdgcn_global_load_lds((gbl_void*)(b_ptr[t] + k0), (lds_void*)(sb + (lw + NL * t) * 1024), 16, 0, 0);
    };
    stage(0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                                // [0] K tile 0 has landed
    for (int kt = 0; kt < nk; ++kt) {
      if (kt + 1 < nk) stage(kt + 1, (kt + 1) & 1);              // into the buffer the MFMA waves left at barrier [kt]
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();                              // [kt + 1]
    }
    return;
  }

  // =========================== MFMA waves ===========================
  const int wr = wave / WC, wc = wave % WC;
  const int r16 = lane & 15, q = lane >> 4;
  f32x4v acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j) acc[i][j] = f32x4v{0.f, 0.f, 0.f, 0.f};
  const int sw = (r16 >> 1) & 7;
  const int a_off = (wr * (BM / WR) + r16) * 128;
  const int b_off = A_BYTES + (wc * (BN / WC) + r16) * 128;
  __builtin_amdgcn_s_barrier();                                  // [0]
  for (int kt = 0; kt < nk; ++kt) {
    const unsigned char* st = smem + (kt & 1) * STAGE_BYTES;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      const int ch = ((4 * ks + q) ^ sw) * 16;
      bf16x8 af[TM], bfr[TN];
#pragma unroll
      for (int i = 0; i < TM; ++i) af[i] = *reinterpret_cast<const bf16x8*>(st + a_off + i * 16 * 128 + ch);
#pragma unroll
      for (int j = 0; j < TN; ++j) bfr[j] = *reinterpret_cast<const bf16x8*>(st + b_off + j * 16 * 128 + ch);
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[j], af[i], acc[i][j], 0, 0, 0);  // C^T tile
    }
    __builtin_amdgcn_s_barrier();                                // [kt + 1]: done with buffer kt & 1; K tile kt + 1 has landed
  }
  // full tiles: hoisted row pointers and 8-byte loads / stores (the same expressions as the generic form, operand for operand)
  const int64_t mw = m0 + wr * (BM / WR), nw = n0 + wc * (BN / WC);
  if (g4_fast_epilogue_ok<EPI, TN, TM>(p, mw, nw)) g4_fast_epilogue<EPI, TN, TM>(p, acc, mw, nw, r16, q);
  else gemm16_epilogue<TM, TN, EPI>(p, acc, mw, nw, r16, q);
}

template <int EPI>
static int launch_conv_wide(GemmParams& p, hipStream_t s) {
  p.tiles_m = (int)((p.M + 255) / 256);
  p.group_m = 8;
  p.tiles_n = (int)(p.N / 192);
  const size_t lds = 2 * (256 + 192) * 128;
  auto kern = conv_wide_kernel<EPI>;
  // (function-local static: initialised once, thread-safe)
  static const bool attr_ok = !(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess);
  if (!attr_ok) {
      lcv_set_error("conv3d: cannot raise dynamic LDS to %zu", lds);
      return LCV_EDEVICE;
  }
  hipLaunchKernelGGL(kern, dim3((unsigned)(p.tiles_m * p.tiles_n)), dim3(1024), lds, s, p);   // 8 MFMA + 8 loader waves
  LCV_LAUNCH_CHECK("conv_wide");
  return LCV_OK;
}
