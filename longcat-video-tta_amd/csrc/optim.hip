// Fused multi-tensor gradient-norm clip + AdamW for the TTA inner loop: one launch over every adapter tensor
// (240 adapters = 480 tensors at qkv+proj) instead of torch's ~10 foreach launches.
//
// The reference trains bf16 adapters with torch.optim.AdamW(foreach) after torch.nn.utils.clip_grad_norm_
// (lora_experiment/scripts/run_lora_tta.py:462-468, 513-514), i.e. every intermediate of the update is ROUNDED
// TO bf16.  The kernel reproduces that op sequence and its rounding points exactly:
//   g  = bf16(g * coef)                                   clip_grad_norm_   (coef is a bf16 value)
//   p  = bf16(p * (1 - lr*wd))                            _foreach_mul_
//   m  = bf16(m + (1-b1) * (g - m))                       _foreach_lerp_    (weight < 0.5 form)
//   v  = bf16(v * b2);  v = bf16(v + ((1-b2) * g) * g)    _foreach_mul_, _foreach_addcmul_
//   d  = bf16(sqrt(v)); d = bf16(d / sqrt(1-b2^t)); d = bf16(d + eps)
//   p  = bf16(p + (-lr/(1-b1^t)) * (m / d))               _foreach_addcdiv_
// For fp32 parameters (the delta methods) the same sequence runs without the bf16 roundings.
// Algorithmic bytes: 14 B / parameter (bf16: p, g, m, v read; p, m, v written).
#include "lcv_common.h"
#include <math.h>

static constexpr int CHUNK = 2048;  // elements per workgroup (256 threads x 8)
// per-tensor sum-of-squares accumulators: chunk c of a tensor adds into slot c % NORM_SLOTS of that tensor's row, the
// coefficient kernel adds the row up.  One slot per tensor makes a 45 M-element weight (22 000 chunks) a queue on one address.
static constexpr int NORM_SLOTS = 64;

__device__ __forceinline__ int find_tensor(const lcv_adam_tensor* t, int n, int64_t chunk) {
  int lo = 0, hi = n - 1;
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (t[mid].first_chunk <= chunk) lo = mid; else hi = mid - 1;
  }
  return lo;
}

template <bool F32>
__global__ __launch_bounds__(256) void grad_sumsq_kernel(const lcv_adam_tensor* __restrict__ tensors, int n,
                                                         float* __restrict__ per_tensor) {
  const int ti = find_tensor(tensors, n, blockIdx.x);
  const lcv_adam_tensor t = tensors[ti];
  const int64_t chunk = (int64_t)blockIdx.x - t.first_chunk;
  const int64_t base = chunk * CHUNK + threadIdx.x * 8;
  float acc = 0.f;
  if (!F32 && base + 8 <= t.numel && (((uintptr_t)t.grad) & 15) == 0) {      // whole 16-byte packet
    float g[8];
    unpack8(*reinterpret_cast<const u16x8*>((const bf16_t*)t.grad + base), g);
#pragma unroll
    for (int e = 0; e < 8; ++e) acc = fmaf(g[e], g[e], acc);
  } else {
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int64_t i = base + e;
      if (i < t.numel) {
        const float g = F32 ? ((const float*)t.grad)[i] : bf2f(((const bf16_t*)t.grad)[i]);
        acc += g * g;
      }
    }
  }
  __shared__ float part[4];
  acc = wave_sum(acc);
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) atomicAdd(per_tensor + (int64_t)ti * NORM_SLOTS + (chunk % NORM_SLOTS), part[0] + part[1] + part[2] + part[3]);
}

// total norm exactly as clip_grad_norm_ composes it: per-tensor norms (rounded to the grad dtype), then the
// norm of those; out[0] = total norm, out[1] = clip coefficient min(max_norm / (total + 1e-6), 1).
template <bool F32>
__global__ __launch_bounds__(256) void clip_coef_kernel(const float* __restrict__ per_tensor, int n, float max_norm,
                                                        float* __restrict__ out) {
  __shared__ float part[4];
  float acc = 0.f;
  for (int i = threadIdx.x; i < n; i += 256) {
    float sumsq = 0.f;
    for (int k = 0; k < NORM_SLOTS; ++k) sumsq += per_tensor[(int64_t)i * NORM_SLOTS + k];
    float nrm = sqrtf(sumsq);
    if (!F32) nrm = bfround(nrm);
    acc += nrm * nrm;
  }
  acc = wave_sum(acc);
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) {
    float total = sqrtf(part[0] + part[1] + part[2] + part[3]);
    float coef;
    if (F32) {
      coef = max_norm / (total + 1e-6f);
    } else {
      total = bfround(total);
      coef = bfround(max_norm / bfround(total + 1e-6f));
    }
    out[0] = total;
    out[1] = fminf(coef, 1.0f);
  }
}

struct AdamScalars {
  float c_wd, w1, b2, c2, bc2_sqrt, eps, step_size;
};

template <bool F32>
__global__ __launch_bounds__(256) void adamw_kernel(const lcv_adam_tensor* __restrict__ tensors, int n,
                                                    const float* __restrict__ clip, const AdamScalars s) {
  const int ti = find_tensor(tensors, n, blockIdx.x);
  const lcv_adam_tensor t = tensors[ti];
  const int64_t base = ((int64_t)blockIdx.x - t.first_chunk) * CHUNK + threadIdx.x * 8;
  const float coef = clip ? clip[1] : 1.0f;
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const int64_t i = base + e;
    if (i >= t.numel) break;
    if (F32) {
      float* P = (float*)t.param; float* M = (float*)t.exp_avg; float* V = (float*)t.exp_avg_sq;
      const float g = __fmul_rn(((const float*)t.grad)[i], coef);
      float p = __fmul_rn(P[i], s.c_wd);
      float m = M[i];
      m = __fadd_rn(m, __fmul_rn(s.w1, __fsub_rn(g, m)));
      float v = __fmul_rn(V[i], s.b2);
      v = __fadd_rn(v, __fmul_rn(__fmul_rn(s.c2, g), g));
      float d = __fadd_rn(__fdiv_rn(__fsqrt_rn(v), s.bc2_sqrt), s.eps);
      p = __fadd_rn(p, __fmul_rn(s.step_size, __fdiv_rn(m, d)));
      P[i] = p; M[i] = m; V[i] = v;
    } else {
      bf16_t* P = (bf16_t*)t.param; bf16_t* M = (bf16_t*)t.exp_avg; bf16_t* V = (bf16_t*)t.exp_avg_sq;
      const float g = bfround(__fmul_rn(bf2f(((const bf16_t*)t.grad)[i]), coef));
      float p = bfround(__fmul_rn(bf2f(P[i]), s.c_wd));
      float m = bf2f(M[i]);
      m = bfround(__fadd_rn(m, __fmul_rn(s.w1, __fsub_rn(g, m))));
      float v = bfround(__fmul_rn(bf2f(V[i]), s.b2));
      v = bfround(__fadd_rn(v, __fmul_rn(__fmul_rn(s.c2, g), g)));
      float d = bfround(__fsqrt_rn(v));
      d = bfround(__fdiv_rn(d, s.bc2_sqrt));
      d = bfround(__fadd_rn(d, s.eps));
      p = bfround(__fadd_rn(p, __fmul_rn(s.step_size, __fdiv_rn(m, d))));
      P[i] = f2bf(p); M[i] = f2bf(m); V[i] = f2bf(v);
    }
  }
}

extern "C" int lcv_grad_norm_clip(const lcv_adam_tensor* tensors, int64_t n_tensors, int64_t total_chunks,
                                  int param_f32, float max_norm, float* per_tensor_ws, float* norm_coef_out,
                                  void* stream) {
  LCV_CHECK_ARG(tensors && per_tensor_ws && norm_coef_out && n_tensors > 0 && total_chunks > 0, "grad_norm_clip: bad arguments");
  hipStream_t s = (hipStream_t)stream;
  if (hipMemsetAsync(per_tensor_ws, 0, sizeof(float) * n_tensors * NORM_SLOTS, s) != hipSuccess) {
    lcv_set_error("grad_norm_clip: memset failed");
    return LCV_EDEVICE;
  }
  if (param_f32) {
    hipLaunchKernelGGL(grad_sumsq_kernel<true>, dim3((unsigned)total_chunks), dim3(256), 0, s, tensors, (int)n_tensors, per_tensor_ws);
    LCV_LAUNCH_CHECK("grad_sumsq");
    hipLaunchKernelGGL(clip_coef_kernel<true>, dim3(1), dim3(256), 0, s, per_tensor_ws, (int)n_tensors, max_norm, norm_coef_out);
  } else {
    hipLaunchKernelGGL(grad_sumsq_kernel<false>, dim3((unsigned)total_chunks), dim3(256), 0, s, tensors, (int)n_tensors, per_tensor_ws);
    LCV_LAUNCH_CHECK("grad_sumsq");
    hipLaunchKernelGGL(clip_coef_kernel<false>, dim3(1), dim3(256), 0, s, per_tensor_ws, (int)n_tensors, max_norm, norm_coef_out);
  }
  LCV_LAUNCH_CHECK("clip_coef");
  return LCV_OK;
}

extern "C" int lcv_adamw_step(const lcv_adam_tensor* tensors, int64_t n_tensors, int64_t total_chunks, int param_f32,
                              const float* norm_coef, double lr, double beta1, double beta2, double eps,
                              double weight_decay, int64_t step, void* stream) {
  LCV_CHECK_ARG(tensors && n_tensors > 0 && total_chunks > 0 && step >= 1, "adamw_step: bad arguments");
  // scalars formed in double exactly as torch/optim/adamw.py does, then narrowed to the kernels' opmath (fp32)
  const double bc1 = 1.0 - pow(beta1, (double)step);
  const double bc2 = 1.0 - pow(beta2, (double)step);
  AdamScalars sc;
  sc.c_wd = (float)(1.0 - lr * weight_decay);
  sc.w1 = (float)(1.0 - beta1);
  sc.b2 = (float)beta2;
  sc.c2 = (float)(1.0 - beta2);
  sc.bc2_sqrt = (float)sqrt(bc2);
  sc.eps = (float)eps;
  sc.step_size = (float)((lr / bc1) * -1.0);
  if (param_f32)
    hipLaunchKernelGGL(adamw_kernel<true>, dim3((unsigned)total_chunks), dim3(256), 0, (hipStream_t)stream, tensors, (int)n_tensors, norm_coef, sc);
  else
    hipLaunchKernelGGL(adamw_kernel<false>, dim3((unsigned)total_chunks), dim3(256), 0, (hipStream_t)stream, tensors, (int)n_tensors, norm_coef, sc);
  LCV_LAUNCH_CHECK("adamw_step");
  return LCV_OK;
}

// ---------------------------------------------------------------------------
// SGD (momentum 0) with weight decay, the default optimizer of full-model TTA
// (lora_experiment/scripts/run_full_tta.py:138-144: SGD(params, lr, momentum=0.0, weight_decay)), in the op order
// and bf16 rounding points of torch.optim.SGD's foreach path after clip_grad_norm_:
//   g = bf16(g * coef);  g = bf16(g + wd * p);  p = bf16(p - lr * g)
// Same descriptor table as AdamW (exp_avg / exp_avg_sq unused).  8 B / parameter read, 2 written.
// ---------------------------------------------------------------------------
template <bool F32>
__global__ __launch_bounds__(256) void sgd_kernel(const lcv_adam_tensor* __restrict__ tensors, int n,
                                                  const float* __restrict__ clip, float lr, float wd) {
  const int ti = find_tensor(tensors, n, blockIdx.x);
  const lcv_adam_tensor t = tensors[ti];
  const int64_t base = ((int64_t)blockIdx.x - t.first_chunk) * CHUNK + threadIdx.x * 8;
  const float coef = clip ? clip[1] : 1.0f;
  if (!F32 && base + 8 <= t.numel && ((((uintptr_t)t.param) | ((uintptr_t)t.grad)) & 15) == 0) {   // whole 16-byte packets
    float p[8], g[8];
    bf16_t* P = (bf16_t*)t.param + base;
    unpack8(*reinterpret_cast<const u16x8*>(P), p);
    unpack8(*reinterpret_cast<const u16x8*>((const bf16_t*)t.grad + base), g);
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      float gg = bfround(__fmul_rn(g[e], coef));
      if (wd != 0.f) gg = bfround(__fadd_rn(gg, __fmul_rn(wd, p[e])));
      p[e] = __fadd_rn(p[e], __fmul_rn(-lr, gg));
    }
    *reinterpret_cast<u16x8*>(P) = pack8(p);
    return;
  }
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const int64_t i = base + e;
    if (i >= t.numel) break;
    if (F32) {
      float* P = (float*)t.param;
      float g = __fmul_rn(((const float*)t.grad)[i], coef);
      g = __fadd_rn(g, __fmul_rn(wd, P[i]));
      P[i] = __fadd_rn(P[i], __fmul_rn(-lr, g));
    } else {
      bf16_t* P = (bf16_t*)t.param;
      const float p = bf2f(P[i]);
      float g = bfround(__fmul_rn(bf2f(((const bf16_t*)t.grad)[i]), coef));
      if (wd != 0.f) g = bfround(__fadd_rn(g, __fmul_rn(wd, p)));
      P[i] = f2bf(__fadd_rn(p, __fmul_rn(-lr, g)));
    }
  }
}

extern "C" int lcv_sgd_step(const lcv_adam_tensor* tensors, int64_t n_tensors, int64_t total_chunks, int param_f32,
                            const float* norm_coef, double lr, double weight_decay, void* stream) {
  LCV_CHECK_ARG(tensors && n_tensors > 0 && total_chunks > 0 && total_chunks <= 0x7fffffff, "sgd_step: bad arguments");
  if (param_f32)
    hipLaunchKernelGGL(sgd_kernel<true>, dim3((unsigned)total_chunks), dim3(256), 0, (hipStream_t)stream, tensors, (int)n_tensors,
                       norm_coef, (float)lr, (float)weight_decay);
  else
    hipLaunchKernelGGL(sgd_kernel<false>, dim3((unsigned)total_chunks), dim3(256), 0, (hipStream_t)stream, tensors, (int)n_tensors,
                       norm_coef, (float)lr, (float)weight_decay);
  LCV_LAUNCH_CHECK("sgd_step");
  return LCV_OK;
}
