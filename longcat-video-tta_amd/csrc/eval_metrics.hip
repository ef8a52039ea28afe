// On-device evaluation of a generated clip against its ground-truth frames: the per-frame squared
// error behind PSNR and the Gaussian-window SSIM the reference gets from torchmetrics
// (delta_experiment/scripts/common.py:663-757, evaluate_generation_metrics + _ssim_single).  The reference pulls the
// frames to the host and loops over them in numpy / torch-CPU; here both metrics are one streaming pass per frame over
// the frames where the decoder left them.
//
// Layout: frames are NHWC [N, H, W, C] with C interleaved, exactly the `[N,H,W,3]` array the pipeline returns; the
// generated clip is fp32 in [0,1], the ground truth either fp32 or raw uint8 (divided by 255 in the kernel, the same
// correctly-rounded fp32 quotient numpy's `/ 255.0 -> float32` gives).  A frame row is treated as ONE flat array of
// W*C floats: the horizontal 11-tap window of channel c at column w is the stride-C stencil j + C*k, so no channel
// logic exists in the kernels and every global access is a contiguous run.
//
// Both kernels are HBM-bound (4 + 1 bytes per element with uint8 ground truth); each writes one fp32 partial per
// workgroup and the caller adds the partials in fp64 (deterministic: no atomics).
#include "lcv_common.h"

namespace {

// uint8 / 255 as the correctly rounded fp32 quotient (what numpy's `/ 255.0` -> float32 gives): the product with 1/255
// in fp64 is within 2^-53 of the exact rational u/255, which never lies that close to an fp32 rounding boundary, so
// rounding it to fp32 equals rounding the exact quotient — 3 instructions instead of the 10 of an IEEE fp32 division.
__device__ __forceinline__ float u8_to_unit(unsigned int u) { return (float)((double)u * (1.0 / 255.0)); }

template <bool GT_U8>
__device__ __forceinline__ float load_gt(const void* gt, int64_t i) {
  if constexpr (GT_U8) return u8_to_unit(((const unsigned char*)gt)[i]);
  else return ((const float*)gt)[i];
}
// the two halves of load_gt, so that a prefetch can hold the raw value and convert after the latency has passed
template <bool GT_U8>
__device__ __forceinline__ unsigned int load_gt_raw(const void* gt, int64_t i) {
  if constexpr (GT_U8) return ((const unsigned char*)gt)[i];
  else return ((const unsigned int*)gt)[i];
}
template <bool GT_U8>
__device__ __forceinline__ float gt_value(unsigned int raw) {
  if constexpr (GT_U8) return u8_to_unit(raw);
  else return __builtin_bit_cast(float, raw);
}

__device__ __forceinline__ float block_sum_256(float v, float* red /*[4]*/) {
  v = wave_sum(v);
  const int wave = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) red[wave] = v;
  __syncthreads();
  return red[0] + red[1] + red[2] + red[3];
}

// ---------------------------------------------------------------------------
// Squared error: partial[n, b] = sum over block b's share of frame n of (gen - gt)^2.  E = H*W*C elements per frame.
// ---------------------------------------------------------------------------
// frames whose element count is not a multiple of 4 (no 16-byte packets): one element per lane per trip
template <bool GT_U8>
__global__ __launch_bounds__(256) void frame_sqerr_scalar_kernel(const float* __restrict__ gen, const void* __restrict__ gt,
                                                                 float* __restrict__ partial, int64_t E) {
  __shared__ float red[4];
  const int64_t n = blockIdx.y;
  float acc = 0.f;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < E; i += (int64_t)gridDim.x * 256) {
    const float d = gen[n * E + i] - load_gt<GT_U8>(gt, n * E + i);
    acc = fmaf(d, d, acc);
  }
  const float s = block_sum_256(acc, red);
  if (threadIdx.x == 0) partial[n * gridDim.x + blockIdx.x] = s;
}

template <bool GT_U8>
__global__ __launch_bounds__(256) void frame_sqerr_kernel(const float* __restrict__ gen, const void* __restrict__ gt,
                                                          float* __restrict__ partial, int64_t E) {
  __shared__ float red[4];
  const int64_t n = blockIdx.y;
  const float* g = gen + n * E;
  const int64_t E4 = E >> 2;                        // 16-byte packets; E % 4 checked on the host
  float acc = 0.f;
  for (int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x; p < E4; p += (int64_t)gridDim.x * 256) {
    const f32x4 a = *(const f32x4*)(g + 4 * p);
    float b[4];
    if constexpr (GT_U8) {
      const unsigned int u = *(const unsigned int*)((const unsigned char*)gt + n * E + 4 * p);
#pragma unroll
      for (int i = 0; i < 4; ++i) b[i] = u8_to_unit((u >> (8 * i)) & 0xffu);
    } else {
      const f32x4 t = *(const f32x4*)((const float*)gt + n * E + 4 * p);
#pragma unroll
      for (int i = 0; i < 4; ++i) b[i] = t[i];
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const float d = a[i] - b[i];
      acc = fmaf(d, d, acc);
    }
  }
  const float s = block_sum_256(acc, red);
  if (threadIdx.x == 0) partial[n * gridDim.x + blockIdx.x] = s;
}

// ---------------------------------------------------------------------------
// SSIM over a separable WIN x WIN window, valid region only.  Two users in the reference:
//   * torchmetrics (TTA runners): 11x11 Gaussian, variances clamped at 0; it pads by 5 (reflect), filters, then crops the
//     same 5 pixels, so the retained map is exactly the windows that lie inside the image ((H-10) x (W-10) per channel);
//   * skimage `structural_similarity` defaults (baseline runner, run_baseline.py:132-136): 7x7 uniform window, sample
//     covariance (factor 49/48), no clamp, border of 3 cropped — again the inside windows only.
//
// A workgroup owns SSIM_COLS consecutive flat output columns j (j = w*C + c over the (W-WIN+1)*C valid ones) and
// SSIM_ROWS output rows, and slides down SSIM_ROWS+WIN-1 input rows.  Per input row it stages SSIM_COLS + (WIN-1)*C
// floats of both images in LDS (double-buffered: one barrier per row), each thread forms the five horizontal window
// sums (x, y, xx, yy, xy) from WIN stride-C taps, and scatters them into WIN running vertical accumulators held in
// registers; the accumulator that has seen WIN rows is a finished window.
// ---------------------------------------------------------------------------
constexpr int SSIM_MAXWIN = 11;
// output rows per workgroup: rows + WIN - 1 input rows is a whole number of WIN-row rounds (44 = 4*11, 42 = 6*7)
constexpr int ssim_rows(int win) { return win == 11 ? 34 : 36; }
constexpr int SSIM_COLS = 512;   // flat output columns per workgroup (2 per thread)
constexpr int SSIM_MAXC = 4;
struct SsimParams {
  float g[SSIM_MAXWIN];
  float c1, c2, cov_norm;
  int clamp_var;
  int H, W, C;
  int Hout, Jout;      // H-(WIN-1), (W-(WIN-1))*C
  int64_t frame_elems;
};

typedef float f32x2_t __attribute__((ext_vector_type(2)));

template <bool GT_U8, int SSIM_WIN>
__global__ __launch_bounds__(256) void frame_ssim_kernel(const float* __restrict__ gen, const void* __restrict__ gt,
                                                         float* __restrict__ partial, SsimParams p) {
  constexpr int SEG = SSIM_COLS + (SSIM_MAXWIN - 1) * SSIM_MAXC;
  __shared__ float xs[2][SEG], ys[2][SEG];
  __shared__ float red[4];
  const int tid = threadIdx.x;
  const int C = p.C;
  const int seg = SSIM_COLS + (SSIM_WIN - 1) * C;
  const int rowlen = p.W * C;
  const int j0 = blockIdx.x * SSIM_COLS;
  constexpr int SSIM_ROWS = ssim_rows(SSIM_WIN);
  const int r0 = blockIdx.y * SSIM_ROWS;
  const int64_t n = blockIdx.z;
  const float* gf = gen + n * p.frame_elems;
  const int rows_out = min(SSIM_ROWS, p.Hout - r0);
  // whole rounds of WIN rows (the unrolled loop below has no early exit); rows past the frame are staged as zeros and
  // the windows they would complete are not counted
  const int rows_in = (rows_out + 2 * SSIM_WIN - 2) / SSIM_WIN * SSIM_WIN;
  // every thread owns TWO flat output columns, 256 apart, held as the halves of a float2 so that the window sums run
  // on the packed fp32 pipe (v_pk_fma_f32): the kernel is VALU-bound (220 flop per output element against 5 bytes)
  const f32x2_t keep = {j0 + tid < p.Jout ? 1.f : 0.f, j0 + tid + 256 < p.Jout ? 1.f : 0.f};

  // next row: global -> registers before the arithmetic of the current row, registers -> LDS after it, so the load
  // latency hides behind ~190 VALU instructions instead of stalling every row
  constexpr int NST = (SEG + 255) / 256;
  float px[NST];
  unsigned int py[NST];
  unsigned int pok = 0;
  auto fetch = [&](int r) {
    const int64_t base = (int64_t)(r0 + r) * rowlen + j0;
    pok = 0;
#pragma unroll
    for (int u = 0; u < NST; ++u) {
      const int i = tid + 256 * u;
      const bool ok = i < seg && j0 + i < rowlen && r0 + r < p.H;
      const int64_t at = ok ? base + i : 0;          // unconditional loads (element 0 of the frame when out of range):
      px[u] = gf[at];                                // no branch, so nothing forces a wait before the arithmetic
      py[u] = load_gt_raw<GT_U8>(gt, n * p.frame_elems + at);
      pok |= (ok ? 1u : 0u) << u;
    }
  };
  auto put = [&](int buf) {
#pragma unroll
    for (int u = 0; u < NST; ++u) {
      const int i = tid + 256 * u;
      const bool ok = (pok >> u) & 1u;
      if (i < seg) { xs[buf][i] = ok ? px[u] : 0.f; ys[buf][i] = ok ? gt_value<GT_U8>(py[u]) : 0.f; }
    }
  };

  const f32x2_t zero = {0.f, 0.f};
  f32x2_t ax[SSIM_WIN], ay[SSIM_WIN], axx[SSIM_WIN], ayy[SSIM_WIN], axy[SSIM_WIN];
#pragma unroll
  for (int i = 0; i < SSIM_WIN; ++i) ax[i] = ay[i] = axx[i] = ayy[i] = axy[i] = zero;
  f32x2_t total = zero;

  fetch(0);
  put(0);
  __syncthreads();
  // The window whose top row is `top` lives in accumulator slot top % WIN; the row loop is unrolled by WIN so that
  // every slot index below is a compile-time constant (no register rotation).
  for (int rb = 0; rb < rows_in; rb += SSIM_WIN) {
#pragma unroll
    for (int U = 0; U < SSIM_WIN; ++U) {
      const int r = rb + U;
      const int buf = r & 1;
      fetch(r + 1);                            // rows past the frame come back as zeros
      f32x2_t hx = zero, hy = zero, hxx = zero, hyy = zero, hxy = zero;
      // the LDS reads of the row in two batches (two round trips instead of WIN), each ahead of its arithmetic
      constexpr int HALF = (SSIM_WIN + 1) / 2;
#pragma unroll
      for (int k0 = 0; k0 < SSIM_WIN; k0 += HALF) {
        f32x2_t xv[HALF], yv[HALF];
#pragma unroll
        for (int k = 0; k < HALF; ++k) {
          if (k0 + k < SSIM_WIN) {
            xv[k] = {xs[buf][tid + (k0 + k) * C], xs[buf][tid + 256 + (k0 + k) * C]};
            yv[k] = {ys[buf][tid + (k0 + k) * C], ys[buf][tid + 256 + (k0 + k) * C]};
          }
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int k = 0; k < HALF; ++k) {
          if (k0 + k < SSIM_WIN) {
            const f32x2_t x = xv[k], y = yv[k];
            const f32x2_t w = {p.g[k0 + k], p.g[k0 + k]};
            hx = w * x + hx;
            hy = w * y + hy;
            hxx = w * (x * x) + hxx;
            hyy = w * (y * y) + hyy;
            hxy = w * (x * y) + hxy;
          }
        }
      }
#pragma unroll
      for (int d = 0; d < SSIM_WIN; ++d) {     // this row is tap d of the window with top row r - d
        const int slot = (U - d + SSIM_WIN) % SSIM_WIN;
        const f32x2_t w = {p.g[d], p.g[d]};
        ax[slot] = w * hx + ax[slot];
        ay[slot] = w * hy + ay[slot];
        axx[slot] = w * hxx + axx[slot];
        ayy[slot] = w * hyy + ayy[slot];
        axy[slot] = w * hxy + axy[slot];
      }
      constexpr int WSLOT_BASE = 1;            // the window that has now seen all WIN rows: top = r - (WIN-1)
      const int done = (U + WSLOT_BASE) % SSIM_WIN;
      if (r >= SSIM_WIN - 1 && r - (SSIM_WIN - 1) < rows_out) {
        const f32x2_t mx = ax[done], my = ay[done];
        f32x2_t sx = p.cov_norm * (axx[done] - mx * mx), sy = p.cov_norm * (ayy[done] - my * my);
        const f32x2_t sxy = p.cov_norm * (axy[done] - mx * my);
        if (p.clamp_var) {
          sx = {fmaxf(sx[0], 0.f), fmaxf(sx[1], 0.f)};
          sy = {fmaxf(sy[0], 0.f), fmaxf(sy[1], 0.f)};
        }
        const f32x2_t num = (2.f * mx * my + p.c1) * (2.f * sxy + p.c2);
        const f32x2_t den = (mx * mx + my * my + p.c1) * (sx + sy + p.c2);
        total += keep * (num / den);
      }
      ax[done] = ay[done] = axx[done] = ayy[done] = axy[done] = zero;   // slot of the window that starts at row r + 1
      put(buf ^ 1);
      __syncthreads();
    }
  }
  const float s = block_sum_256(total[0] + total[1], red);
  if (tid == 0) partial[(n * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x] = s;
}

}  // namespace

static int64_t sqerr_num_partials(int64_t E) {
  int64_t b = (E / 4 + 255) / 256;
  return b < 1 ? 1 : (b > 64 ? 64 : b);
}
static int64_t ssim_num_partials(int64_t H, int64_t W, int64_t C, int64_t win) {
  if (H < win || W < win || C < 1) return 0;
  const int64_t gx = ((W - (win - 1)) * C + SSIM_COLS - 1) / SSIM_COLS, gy = (H - (win - 1) + ssim_rows((int)win) - 1) / ssim_rows((int)win);
  return gx * gy;
}

extern "C" int lcv_frame_metric_partials(int64_t H, int64_t W, int64_t C, int win, int64_t* n_sqerr, int64_t* n_ssim) {
  LCV_CHECK_ARG(n_sqerr && n_ssim, "frame_metric_partials: null pointer");
  LCV_CHECK_ARG(H > 0 && W > 0 && C > 0, "frame_metric_partials: bad frame size");
  *n_sqerr = sqerr_num_partials(H * W * C);
  *n_ssim = ssim_num_partials(H, W, C, win);
  return LCV_OK;
}

extern "C" int lcv_frame_sqerr(const float* gen, const void* gt, int gt_is_u8, float* partials, int64_t N, int64_t E,
                               void* stream) {
  LCV_CHECK_ARG(gen && gt && partials, "frame_sqerr: null pointer");
  LCV_CHECK_ARG(N > 0 && N <= 65535 && E > 0, "frame_sqerr: N=%ld frames of E=%ld elements", (long)N, (long)E);
  const dim3 grid((unsigned)sqerr_num_partials(E), (unsigned)N);
  if (E % 4 != 0) {
    if (gt_is_u8)
      hipLaunchKernelGGL(frame_sqerr_scalar_kernel<true>, grid, dim3(256), 0, (hipStream_t)stream, gen, gt, partials, E);
    else
      hipLaunchKernelGGL(frame_sqerr_scalar_kernel<false>, grid, dim3(256), 0, (hipStream_t)stream, gen, gt, partials, E);
  } else if (gt_is_u8)
    hipLaunchKernelGGL(frame_sqerr_kernel<true>, grid, dim3(256), 0, (hipStream_t)stream, gen, gt, partials, E);
  else
    hipLaunchKernelGGL(frame_sqerr_kernel<false>, grid, dim3(256), 0, (hipStream_t)stream, gen, gt, partials, E);
  LCV_LAUNCH_CHECK("frame_sqerr");
  return LCV_OK;
}

template <int WIN>
static void launch_ssim(const float* gen, const void* gt, int gt_is_u8, float* partials, int64_t N, const SsimParams& p,
                        hipStream_t stream) {
  const dim3 grid((unsigned)((p.Jout + SSIM_COLS - 1) / SSIM_COLS), (unsigned)((p.Hout + ssim_rows(WIN) - 1) / ssim_rows(WIN)), (unsigned)N);
  if (gt_is_u8)
    hipLaunchKernelGGL((frame_ssim_kernel<true, WIN>), grid, dim3(256), 0, stream, gen, gt, partials, p);
  else
    hipLaunchKernelGGL((frame_ssim_kernel<false, WIN>), grid, dim3(256), 0, stream, gen, gt, partials, p);
}

extern "C" int lcv_frame_ssim(const float* gen, const void* gt, int gt_is_u8, float* partials, int64_t N, int64_t H,
                              int64_t W, int64_t C, const float* window, int win, float cov_norm, int clamp_var,
                              float c1, float c2, void* stream) {
  LCV_CHECK_ARG(gen && gt && partials && window, "frame_ssim: null pointer");
  LCV_CHECK_ARG(win == 7 || win == 11, "frame_ssim: window %d (7 = skimage default, 11 = torchmetrics default)", win);
  LCV_CHECK_ARG(C >= 1 && C <= SSIM_MAXC, "frame_ssim: C=%ld channels (1..%d interleaved)", (long)C, SSIM_MAXC);
  LCV_CHECK_ARG(H >= win && W >= win && H < (1 << 20) && W * C < (1 << 24),
                "frame_ssim: %ldx%ld frame is smaller than the window (or absurdly large)", (long)H, (long)W);
  LCV_CHECK_ARG(N > 0 && N <= 65535, "frame_ssim: N=%ld frames", (long)N);
  SsimParams p;
  for (int i = 0; i < SSIM_MAXWIN; ++i) p.g[i] = i < win ? window[i] : 0.f;
  p.c1 = c1; p.c2 = c2; p.cov_norm = cov_norm; p.clamp_var = clamp_var;
  p.H = (int)H; p.W = (int)W; p.C = (int)C;
  p.Hout = (int)(H - (win - 1));
  p.Jout = (int)((W - (win - 1)) * C);
  p.frame_elems = H * W * C;
  if (win == 11) launch_ssim<11>(gen, gt, gt_is_u8, partials, N, p, (hipStream_t)stream);
  else launch_ssim<7>(gen, gt, gt_is_u8, partials, N, p, (hipStream_t)stream);
  LCV_LAUNCH_CHECK("frame_ssim");
  return LCV_OK;
}
