/*
 * lcv_hip.h — C ABI of liblcv_hip.so: the MI355X (gfx950) kernels under the
 * LongCat-Video denoise-and-adapt hot path.
 *
 * The reference (FifthEpoch/longcat-video-tta) has no FFI of its own: its
 * boundary with this path is the Python object protocol of the un-vendored
 * `longcat_video` package.  This header is the boundary
 * *under* that protocol: every entry point below replaces a third-party GPU
 * kernel the reference reaches through torch / flash-attn, and cites the
 * reference call site (paths relative to the reference tree) that drives it.
 *
 * Conventions
 *   - plain pointers + sizes, no torch types; all pointers are DEVICE pointers
 *     unless a name ends in _host;
 *   - bf16 tensors are passed as `const void*` / `void*` (raw 16-bit storage);
 *   - `stream` is a hipStream_t passed as void* (NULL = default stream);
 *   - return 0 on success, a negative LCV_E* code otherwise; the message is
 *     retrievable with lcv_last_error() (thread-local);
 *   - no allocation, no ownership transfer, re-entrant per stream: any
 *     workspace is supplied by the caller;
 *   - strides are in ELEMENTS.
 */
#ifndef LCV_HIP_H
#define LCV_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define LCV_OK 0
#define LCV_EINVAL (-1)   /* bad shape / alignment / argument */
#define LCV_EDEVICE (-2)  /* no gfx950 device or HIP runtime error */
#define LCV_ELAUNCH (-3)  /* kernel launch failed */

/* ---- library -------------------------------------------------------- */
int lcv_version(void);                 /* ABI version (monotonic integer) */
const char* lcv_last_error(void);      /* message of the last failure on this thread */
/* 0 when the current device is gfx950; LCV_EDEVICE otherwise. */
int lcv_device_check(void);

/* ---- A/B knobs -------------------------------------------------------
 * Environment variables the launchers consult to pick between SHIPPED alternatives of a kernel (same results, bit for bit unless
 * a line says otherwise; defaults are the measured-fastest forms).  They are read ONCE, at the library's first use;
 * lcv_knobs_reload() reads the environment again (tests and A/B harnesses that flip a knob inside one process; returns the
 * number of knobs that are set, never an error); lcv_knobs_list() returns the names below, space separated.  Nothing else in the
 * environment is looked at.
 *   LCV_GEMM_TILE          k = gemm4k.h (default where its one epilogue serves the call), 9 / 8 = the 8-phase kernel persistent /
 *                          per-tile, 6 / 7 = one-barrier 256 / 128 tiles (the bit-level reference of the tests), 2 / 1 = the same
 *                          tiles on the 32x32x16 MFMA
 *   LCV_GEMM_GROUP_M       tile rows per group of the XCD-contiguous tile order (default 3 for gemm4k, 6 for the 8-phase kernel)
 *   LCV_GEMM_FAST_EPI      0 = the 8-phase kernel takes the generic epilogue on interior tiles too
 *   LCV_GEMM_SPLITK_TAIL   0 = the 8-phase kernel does not split a thin last round of tiles along K (summation order differs)
 *   LCV_CONV_8P, LCV_CONV_N192, LCV_CONV_ROWS, LCV_CONV_ROWS_GRID, LCV_CONV_ROWS_ORDER
 *                          which convolution kernel / tile order the VAE stages take (csrc/conv_rows.h, conv_wide.h)
 *   LCV_ATTN_FWD_W64       0 = the two-waves-per-SIMD forward of round 2 instead of attn_fwd_w64_kernel (cross-check; 1.4e-5 apart)
 *   LCV_ATTN_XCD           0 = no head-per-XCD workgroup order in the forward
 *   LCV_ATTN_BWD_VAR       bit 0 / bit 1 = second-form pass B / pass A of the backward (default 3; 0 = the general-scale forms)
 *   LCV_ATTN_BWD_DKV_WAVES, LCV_ATTN_BWD_DQ_WAVES   8 = one 8-wave workgroup per CU instead of two 4-wave ones
 *   LCV_ATTN_BWD_PIPE, LCV_ATTN_BWD_STAGGER, LCV_ATTN_BWD_XCD   schedule variants of the backward passes measured and left off */
int lcv_knobs_reload(void);
const char* lcv_knobs_list(void);

/* ---- AdaLN modulate / LayerNorm ------------------------------------- */
/* y = LN_noaffine_fp32(x) * (1 + scale[b,t,:]) + shift[b,t,:]  -> bf16
 * x,y: [B, T*S, C] bf16 (S tokens per latent frame); shift/scale: fp32 rows of
 * a [B, T, mod_stride] table (the adaLN_modulation output), addressed as
 * mod + (b*T+t)*mod_stride + {shift_off,scale_off}.
 * Replaces upstream modulate_fp32(mod_norm_*, x, shift, scale) reached from the
 * block forward; layout evidence: delta_experiment/scripts/run_film_tta.py:5-12,134-141. */
int lcv_adaln_modulate_fwd(const void* x, const float* mod, void* y,
                           int64_t B, int64_t T, int64_t S, int64_t C,
                           int64_t mod_stride, int64_t shift_off, int64_t scale_off,
                           float eps, void* stream);
/* Backward of the above w.r.t. x and (optionally) the modulation table.
 * dx: bf16 [B,T*S,C]; dmod (nullable): fp32 [B,T,mod_stride], ACCUMULATED at
 * shift_off / scale_off (caller zero-fills once per block). */
int lcv_adaln_modulate_bwd(const void* x, const float* mod, const void* dy,
                           void* dx, float* dmod,
                           int64_t B, int64_t T, int64_t S, int64_t C,
                           int64_t mod_stride, int64_t shift_off, int64_t scale_off,
                           float eps, const void* dres, void* stream);
/* dres (nullable, bf16 [B,T*S,C]): a second gradient w.r.t. x — the one that arrives through the residual path of the same
 * block — added to dx in fp32 before the one bf16 rounding.  It replaces the separate elementwise add autograd would launch
 * for the fork x -> {norm, residual} (three per block and step in the TTA inner loop, run_lora_tta.py:512). */
/* y = LN_fp32(x) * w + b  (affine LayerNorm_FP32; pre_crs_attn_norm).
 * Module name evidence: delta_experiment/scripts/run_norm_tune_tta.py:78-96. */
int lcv_layernorm_affine_fwd(const void* x, const float* w, const float* b, void* y,
                             int64_t rows, int64_t C, float eps, void* stream);
/* dx (bf16) and, if non-NULL, dw/db fp32 [C] ACCUMULATED (atomics). */
int lcv_layernorm_affine_bwd(const void* x, const float* w, const void* dy,
                             void* dx, float* dw, float* db,
                             int64_t rows, int64_t C, float eps, const void* dres, void* stream);

/* ---- gated residual ------------------------------------------------- */
/* out = bf16( f32(x) + gate[b,t,:] * f32(y) ); gate addressed like shift above.
 * gate == NULL means gate = 1 (plain residual add, cross-attention branch). */
int lcv_gate_residual_fwd(const void* x, const void* y, const float* mod, void* out,
                          int64_t B, int64_t T, int64_t S, int64_t C,
                          int64_t mod_stride, int64_t gate_off, void* stream);
/* dy = gate*dout (bf16); dmod[gate_off] += sum_tokens(dout*y) (nullable). dx = dout (alias, not written). */
int lcv_gate_residual_bwd(const void* y, const float* mod, const void* dout,
                          void* dy, float* dmod,
                          int64_t B, int64_t T, int64_t S, int64_t C,
                          int64_t mod_stride, int64_t gate_off, void* stream);

/* ---- q/k RMSNorm + 3-D RoPE ----------------------------------------- */
/* For every token n and head h: q <- rope(rmsnorm(q)*wq), k <- rope(rmsnorm(k)*wk),
 * v copied (skipped when v_out == NULL or v_out == v_in).
 * Inputs are addressed as ptr + b*sb + n*sn + h*D (+d); D == 128.
 * cs: [N_pos, D/2] float2 (cos, sin) table, row = pos_off + n (global position, so a
 * sequence-parallel shard passes its own offset); cs == NULL skips RoPE
 * (text cross-attention q/k norm).  RoPE pairs are interleaved (2i, 2i+1).
 * q_scale (> 0, 1 = none) multiplies the q output before its single bf16 rounding: the self-attention
 * path passes head_dim^-0.5 * log2(e) here and scale = ln 2 to lcv_attn_fwd/bwd, whose exponent is
 * then q.k itself (same softmax, one multiply-subtract per score less).
 * Replaces upstream q_norm/k_norm/rope_3d inside Attention.forward
 * (attribute names: lora_experiment/scripts/run_lora_tta.py:142-168). */
int lcv_qknorm_rope_fwd(const void* q_in, const void* k_in, const void* v_in,
                        void* q_out, void* k_out, void* v_out,
                        const void* wq, const void* wk, const void* cs,
                        int64_t B, int64_t N, int64_t H,
                        int64_t in_sb, int64_t in_sn,       /* q_in/k_in/v_in strides */
                        int64_t q_sb, int64_t q_sn,         /* q_out strides */
                        int64_t kv_sb, int64_t kv_sn,       /* k_out/v_out strides */
                        int64_t pos_off, float eps, float q_scale, void* stream);
/* Backward: given dq_out, dk_out (same addressing as q_out/k_out) and the
 * pre-norm q_in/k_in, writes dq_in, dk_in (strides din_*); q_scale as in the forward.  dwq / dwk
 * (nullable, fp32 [dw_slots, 128], ACCUMULATED into by atomics - zero them first; token n adds into row
 * n % dw_slots, the caller sums the rows) receive the gradients of the two norm weights (norm-weight tuning,
 * delta_experiment/scripts/run_norm_tune_tta.py:87-98; full-model TTA); the dq share is w.r.t. the weight as it
 * enters the forward, i.e. it carries q_scale. */
int lcv_qknorm_rope_bwd(const void* q_in, const void* k_in,
                        const void* dq_out, const void* dk_out,
                        void* dq_in, void* dk_in,
                        const void* wq, const void* wk, const void* cs,
                        int64_t B, int64_t N, int64_t H,
                        int64_t in_sb, int64_t in_sn,
                        int64_t q_sb, int64_t q_sn,
                        int64_t kv_sb, int64_t kv_sn,
                        int64_t din_sb, int64_t din_sn,
                        int64_t pos_off, float eps, float q_scale, float* dwq, float* dwk, int64_t dw_slots, void* stream);

/* Sinusoidal timestep features, fp32: out[i, :dim/2] = cos(t_i f), out[i, dim/2:] = sin(t_i f), f_j = max_period^(-j/(dim/2)).
 * Input of the DiT's timestep MLP (upstream TimestepEmbedder.timestep_embedding; outer forward restated at
 * delta_experiment/scripts/run_delta_a.py:160-166). */
int lcv_timestep_embedding(const float* t, float* out, int64_t n, int64_t dim, float max_period, void* stream);

/* ---- flash attention (dense, non-causal, head_dim 128, bf16, fp32 acc) */
/* o[b,n,h,:] = softmax(q k^T * scale) v ; q: Nq rows, k/v: Nk rows.
 * Every tensor is addressed ptr + b*s_b + n*s_n + h*s_h + d with d contiguous.
 * lse (nullable): fp32 [B, H, Nq] natural-log-sum-exp of the scaled scores.
 * Replaces flash_attn_func / flash_attn_varlen_func requested by
 * enable_flashattn2=True (delta_experiment/scripts/common.py:73); the
 * conditioning split (cond-q x cond-kv, noise-q x all-kv) and the KV-cached
 * denoise step are expressed by the caller through pointer offsets. */
int lcv_attn_fwd(const void* q, const void* k, const void* v, void* o, float* lse,
                 int64_t B, int64_t H, int64_t Nq, int64_t Nk,
                 int64_t q_sb, int64_t q_sn, int64_t q_sh,
                 int64_t k_sb, int64_t k_sn, int64_t k_sh,
                 int64_t v_sb, int64_t v_sn, int64_t v_sh,
                 int64_t o_sb, int64_t o_sn, int64_t o_sh,
                 float scale, void* stream);
/* Which kernel the most recent lcv_attn_fwd call of this thread launched (a static string, never NULL; "none" before the
 * first call): the library picks the body by shape and scale (software-pipelined self-attention body, phase-ordered body,
 * short-key cross-attention body), and bench.py labels its roofline object with what actually ran. */
const char* lcv_attn_fwd_last_kernel(void);
/* Backward (two passes, no atomics, bit-reproducible: dK/dV per 128-key workgroup, dQ per 256-query workgroup; see
 * csrc/attn_bwd.hip).  d_o shares o's strides.  delta_ws: fp32 workspace of lcv_attn_bwd_ws_floats(B, H, Nq, Nk) floats:
 * B*H*(Nq + 2*roundup(Nq, 32)) (delta, then the padded -lse*log2(e) / -delta rows the second-form pass A streams into LDS),
 * plus qsplit*B*H*Nk*256 when the query sweep of a short-key call (Nk <= 128: the 77-key text cross-attention) is split over
 * workgroups - one fp32 dK / dV slice per split, added in split order by a finishing kernel (the size is an UPPER bound: the
 * slices are counted for every call of such sizes, also the unit-scale ones whose pass A does not use them).  accumulate_kv != 0 adds into the
 * existing dk/dv (second region of the conditioning split).  dq/dk/dv are addressed like q/k/v with their own strides. */
int64_t lcv_attn_bwd_ws_floats(int64_t B, int64_t H, int64_t Nq, int64_t Nk);   /* host-only; a size, not a status */
int lcv_attn_bwd(const void* q, const void* k, const void* v, const void* o,
                 const void* d_o, const float* lse,
                 void* dq, void* dk, void* dv,
                 float* delta_ws, int accumulate_kv,
                 int64_t B, int64_t H, int64_t Nq, int64_t Nk,
                 int64_t q_sb, int64_t q_sn, int64_t q_sh,
                 int64_t k_sb, int64_t k_sn, int64_t k_sh,
                 int64_t v_sb, int64_t v_sn, int64_t v_sh,
                 int64_t o_sb, int64_t o_sn, int64_t o_sh,
                 int64_t dq_sb, int64_t dq_sn, int64_t dq_sh,
                 int64_t dk_sb, int64_t dk_sn, int64_t dk_sh,
                 int64_t dv_sb, int64_t dv_sn, int64_t dv_sh,
                 float scale, void* stream);

/* ---- projection GEMM (+ bias, + fused LoRA rank-r term, + epilogues) -- */
#define LCV_EPI_NONE 0        /* c = acc + bias */
#define LCV_EPI_SWIGLU 1      /* w rows interleaved [32 gate | 32 up]; c[M, N/2] = silu(g)*u.  A non-NULL `resid` is an OUTPUT here:
                               * bf16 [M, N], the pre-activation (gate | up) rows in the weight's column order, kept for
                               * lcv_swiglu_bwd_interleaved (training) */
#define LCV_EPI_GATE_RESIDUAL 2 /* c = resid + gate[b,t,:]*(acc+bias) (fp32), rows are tokens */
#define LCV_EPI_GELU_TANH 3   /* c = gelu_tanh(acc + bias) */
#define LCV_EPI_SILU 4        /* c = silu(acc + bias) */
/* c[M,N] = a[M,K] @ w[N,K]^T (+ bias[N]) (+ a2[M,K2] @ w2[N,K2]^T), bf16 in,
 * fp32 accumulate, bf16 (out_f32 == 0) or fp32 out.  K % 64 == 0, K2 % 64 == 0
 * (K2 = LoRA rank zero-padded to 64, a2 = s*(x A^T), w2 = B).  lda/ldw/ldc
 * are row strides.  Replaces cuBLAS via nn.Linear for attn.qkv / attn.proj /
 * cross_attn.{q_linear,kv_linear,proj} / ffn.w{1,2,3} / adaLN_modulation and the
 * LoRALinear forward  (lora_experiment/scripts/run_lora_tta.py:224-260). */
int lcv_gemm_nt(const void* a, const void* w, const void* bias,
                const void* a2, const void* w2,
                void* c, int64_t M, int64_t N, int64_t K, int64_t K2,
                int64_t lda, int64_t ldw, int64_t lda2, int64_t ldw2, int64_t ldc,
                int epilogue, int out_f32,
                const void* resid, const float* mod, int64_t rows_per_frame,
                int64_t mod_stride, int64_t gate_off,
                void* stream);
/* Optional fp32 workspace (>= 16-byte aligned device memory, e.g. 64 MiB; NULL / 0 to withdraw) for the split-K tail of
 * lcv_gemm_nt's persistent kernel: when the tile count leaves a last round with few busy CUs (784 tiles on 256 CUs at
 * the reference's 480p operating point), those tail tiles are split along K into one partial round, summed and finished
 * by a reduce kernel.  The library never allocates: without a workspace the tail runs unsplit.  One workspace per
 * process, used by one stream at a time. */
int lcv_gemm_set_workspace(void* ws, int64_t bytes);
/* LoRA down-projection: h[M, Rpad] = bf16( s * bf16(x[M,K] @ A[R,K]^T) ), columns R..Rpad-1 zero.
 * (lora_down of LoRALinear, run_lora_tta.py:247-260). */
int lcv_lora_down(const void* x, const void* A, void* h, int64_t M, int64_t K, int64_t R,
                  int64_t Rpad, int64_t ldx, float s, void* stream);
/* LoRA-only backward (no dW of the frozen base weight), with h = bf16(x A^T), g = s * dy B (= lcv_lora_down(dy, B^T)):
 *   dx = dy W + g A            -> lcv_gemm_nt(dy, W^T, a2 = g, w2 = A^T)
 *   dA[R,K]   = g^T x          -> lcv_tn_skinny(g, x)
 *   dB^T[R,N] = s * h^T dy     -> lcv_tn_skinny(h, dy, scale = s)
 * out[r,k] = scale * sum_m g[m,r] x[m,k]; g: [M,Rpad] bf16, x: [M,K] bf16 (row stride ldx), out: [R,K] fp32.
 * ws (16-byte aligned, >= lcv_tn_skinny_ws_bytes(M, K, R)): per-row-group partial sums + a fixed-order reduction, `out` is
 * overwritten (bit-reproducible).  ws == NULL: accumulated into `out` with fp32 atomics (caller zero-fills). */
int lcv_tn_skinny(const void* g, const void* x, float* out, int64_t M, int64_t K, int64_t R, int64_t Rpad,
                  int64_t ldx, float scale, float* ws, int64_t ws_bytes, void* stream);
int64_t lcv_tn_skinny_ws_bytes(int64_t M, int64_t K, int64_t R);
/* fp32 islands (t_embedder MLP, adaLN_modulation under the upstream fp32 autocast; run_delta_a.py:161-165):
 * out[M,N] fp32 = act_in(a[M,K] fp32) @ w[N,K]^T (bf16 weights widened) + bias;  act_in: 0 none, 1 SiLU. */
int lcv_linear_f32_smallm(const float* a, const void* w, const void* bias, float* out, int64_t M, int64_t N,
                          int64_t K, int act_in, void* stream);
/* da[M,K] fp32 = act_in'(a) * (dy[M,N] @ w[N,K])  (gradient towards the timestep embedding: delta-A/B TTA). */
int lcv_linear_f32_smallm_bwd(const float* dy, const void* w, const float* a, float* da, int64_t M, int64_t N,
                              int64_t K, int act_in, void* stream);

/* ---- SwiGLU --------------------------------------------------------- */
int lcv_swiglu_fwd(const void* gate, const void* up, void* out, int64_t rows, int64_t F,
                   int64_t ld_in, void* stream);
int lcv_swiglu_bwd(const void* gate, const void* up, const void* dout, void* dgate, void* dup,
                   int64_t rows, int64_t F, int64_t ld_in, void* stream);
/* The backward on the fused layout: gu [rows, 2F] = per 64 columns 32 gate values then their 32 up partners (what
 * lcv_gemm_nt writes through `resid` under LCV_EPI_SWIGLU); dgu [rows, 2F] gets the gradients in the same layout. */
int lcv_swiglu_bwd_interleaved(const void* gu, const void* dout, void* dgu, int64_t rows, int64_t F, void* stream);

/* ---- patchify / unpatchify ------------------------------------------ */
/* x [B,Cin,T,H,W] bf16 -> tokens [B, T*(H/2)*(W/2), Kpad] bf16, k = c*4 + ph*2 + pw (conv3d weight
 * flattening order of x_embedder.proj, patch (1,2,2)); columns >= Cin*4 zero. */
int lcv_patchify(const void* x, void* tok, int64_t B, int64_t Cin, int64_t T, int64_t H, int64_t W,
                 int64_t Kpad, void* stream);
/* tokens [B, N, 4*Cout] (ph, pw, c order; bf16 or fp32) -> out [B,Cout,T,H,W] fp32
 * (upstream unpatchify; delta_experiment/scripts/run_delta_a.py:213-217). */
int lcv_unpatchify(const void* tok, float* out, int64_t B, int64_t Cout, int64_t T, int64_t H, int64_t W,
                   int tok_is_f32, void* stream);
/* dtok [B, N, 4*Cout] fp32 <- dout [B,Cout,T,H,W] fp32. */
int lcv_unpatchify_bwd(const float* dout, float* dtok, int64_t B, int64_t Cout, int64_t T, int64_t H, int64_t W,
                       void* stream);

/* ---- denoise step glue ---------------------------------------------- */
/* CFG-zero-star combine + sign + Euler update, fp32:
 *   st = <c,u>/(<u,u>+1e-8) per sample (computed by the callee from the partial sums in ws),
 *   v  = u*st + g*(c - u*st);  x <- x + dt * (negate ? -v : v)
 * cond/uncond: fp32 [B, n]; x: fp32 [B, n] in place; ws: fp32 [B, 256, 2] workspace (per-slice partial sums, added in a
 * fixed order: the step is bit-reproducible). */
int lcv_cfg_euler_step(const float* cond, const float* uncond, float* x, float* ws,
                       int64_t B, int64_t n, float guidance, float dt, int negate,
                       int use_zero_star, void* stream);
/* x <- x + dt*v (no CFG). */
int lcv_euler_step(const float* v, float* x, int64_t n, float dt, int negate, void* stream);

/* ---- flow-matching loss pieces --------------------------------------- */
/* noisy = (1-sigma)*x0 + sigma*eps (bf16 out), per-sample sigma fp32 [B].
 * delta_experiment/scripts/common.py:458-466. */
int lcv_fm_noise(const void* x0, const void* eps, const float* sigma, void* out,
                 int64_t B, int64_t per_sample, void* stream);
/* loss = mean( (pred[:,:,Tc:] - (eps - x0))^2 ) fp32, and dpred = 2/n * diff on the
 * target slice, 0 on the cond slice.  pred fp32 [B,C,T,HW]; eps/x0 bf16 [B,C,Tt,HW].
 * common.py:485-488.  loss_out: fp32 [1]; dpred nullable; ws: fp32 [LCV_FM_MSE_BLOCKS] scratch - the per-workgroup partial sums
 * are added in a fixed order by a second launch (no atomics: the training loss is the same float on every run). */
#define LCV_FM_MSE_BLOCKS 1024
int lcv_fm_mse(const float* pred, const void* eps, const void* x0, float* loss_out, float* dpred, float* ws,
               int64_t B, int64_t C, int64_t T, int64_t Tc, int64_t HW, void* stream);
/* Per-sample, no-gradient, deterministic form: loss_out[b] = mean over sample b's target slice.  One launch pair scores
 * the early stopper's whole anchor set (sigmas x noise draws batched into ONE forward) where the reference runs one forward
 * and one `.item()` per (sigma, draw): delta_experiment/scripts/common.py:492-559 (the loop at :530-557),
 * early_stopping.py:296-317.  eps / x0: bf16 [.., C, Tt, HW], sample b at element offset b * {eps,x0}_bstride
 * (0 = one tensor shared by all samples).  ws: fp32 [B * LCV_FM_MSE_PARTS] scratch; loss_out: fp32 [B]. */
#define LCV_FM_MSE_PARTS 256
int lcv_fm_mse_samples(const float* pred, const void* eps, const void* x0, float* loss_out, float* ws,
                       int64_t B, int64_t C, int64_t T, int64_t Tc, int64_t HW, int64_t eps_bstride,
                       int64_t x0_bstride, void* stream);

/* ---- fused multi-tensor AdamW + global-norm clip --------------------- */
/* One descriptor per parameter tensor (device array).  Tensors are cut into 2048-element chunks; a tensor's
 * chunks are [first_chunk, first_chunk + ceil(numel/2048)). */
typedef struct {
  void* param;      /* bf16 or fp32 (see param_f32) */
  void* grad;       /* same dtype as param */
  void* exp_avg;    /* same dtype as param */
  void* exp_avg_sq; /* same dtype as param */
  int64_t numel;
  int64_t first_chunk;
} lcv_adam_tensor;
/* torch.nn.utils.clip_grad_norm_: per-tensor norms (rounded to the grad dtype) -> total norm -> coefficient
 * min(max_norm / (total + 1e-6), 1).  per_tensor_ws: fp32 [n_tensors, 64] (64 partial sums of squares per tensor);
 * norm_coef_out: fp32 [2] = {norm, coef}.
 * lora_experiment/scripts/run_lora_tta.py:513. */
int lcv_grad_norm_clip(const lcv_adam_tensor* tensors, int64_t n_tensors, int64_t total_chunks, int param_f32,
                       float max_norm, float* per_tensor_ws, float* norm_coef_out, void* stream);
/* torch.optim.AdamW(foreach).step in one launch, with the bf16 rounding points of the foreach op sequence;
 * applies norm_coef[1] to the gradients on the fly when norm_coef != NULL.  run_lora_tta.py:462-468, 514. */
int lcv_adamw_step(const lcv_adam_tensor* tensors, int64_t n_tensors, int64_t total_chunks, int param_f32,
                   const float* norm_coef, double lr, double beta1, double beta2, double eps, double weight_decay,
                   int64_t step, void* stream);

/* torch.optim.SGD(momentum=0, weight_decay) after clip_grad_norm_, foreach op order and bf16 rounding points:
 * g = g*coef; g += wd*p; p -= lr*g.  Same descriptor table as AdamW (moment pointers unused).  The default optimizer
 * of full-model TTA: lora_experiment/scripts/run_full_tta.py:138-144, 179-180. */
int lcv_sgd_step(const lcv_adam_tensor* tensors, int64_t n_tensors, int64_t total_chunks, int param_f32,
                 const float* norm_coef, double lr, double weight_decay, void* stream);

/* ---- dense backward pieces of full-model TTA (run_full_tta.py:95-215: loss.backward() over ALL DiT parameters) ----
 * lcv_transpose_pad: out[N, Mpad] = in[M, N]^T (row stride ld), columns >= M zero, Mpad % 64 == 0.  With it a dense
 *   weight gradient dW[N,K] = dY^T . X is lcv_gemm_nt(A = dY^T [N, Mpad], W = X^T [K, Mpad]).
 * lcv_rowsum: out[r] = sum_c in[r, c] (bias gradient from dY^T); out bf16 or fp32.
 * lcv_linear_f32_smallm_wgrad: weight / bias gradients of the fp32-island linears (adaLN modulation, timestep MLP):
 *   dw[n,k] = sum_m dy[m,n] act(a[m,k]), db[n] = sum_m dy[m,n], bf16 outputs, act_in as lcv_linear_f32_smallm.
 * lcv_gelu_tanh_fwd / _bwd: the caption embedder's activation when its linears are trainable (unfused form). */
int lcv_transpose_pad(const void* in, void* out, int64_t M, int64_t N, int64_t ld, int64_t Mpad, void* stream);
int lcv_rowsum(const void* in, void* out, int64_t rows, int64_t cols, int out_f32, void* stream);
int lcv_linear_f32_smallm_wgrad(const float* dy, const float* a, void* dw, void* db, int64_t M, int64_t N, int64_t K,
                                int act_in, void* stream);
int lcv_gelu_tanh_fwd(const void* x, void* y, int64_t n, void* stream);
int lcv_gelu_tanh_bwd(const void* x, const void* dy, void* dx, int64_t n, void* stream);

/* ---- VAE stages (WAN-style causal 3-D conv VAE; upstream AutoencoderKLWan.decode / .encode, contract at
 * delta_experiment/scripts/common.py:209-221) --------------------------------------------------------------- */
/* Causal conv3d as an implicit GEMM on the MFMA core.  x [B,Tin,Hin,Win,ldx] channels-last bf16 with Cin % 32 == 0 valid
 * channels and pixel stride ldx = Cin rounded up to a multiple of 64 (padding channels zero, in x and in w - a caller may as
 * well pass Cin = ldx); w [Cout, kt*kh*kw*ldx] with K ordered (dt,dh,dw,cin); out [B,T,H,W,ldc], (H,W) doubled when
 * up2x (nearest upsample folded into the gather).  kt-1 zero frames of causal padding in front, zero spatial padding.
 * resid (nullable, laid out like out): out = resid + bf16(conv + bias).  zero_page: >= 128 bytes of device zeros. */
int lcv_causal_conv3d(const void* x, const void* w, const void* bias, const void* resid, void* out,
                      const void* zero_page, int64_t B, int64_t Tin, int64_t Hin, int64_t Win, int64_t Cin,
                      int64_t Cout, int64_t ldc, int kt, int kh, int kw, int up2x, void* stream);
/* Which kernel the most recent lcv_causal_conv3d / lcv_conv3d_strided call of this thread launched (static string, "none"
 * before the first call): the row-tile kernel for the 96-channel stages or the implicit GEMM (csrc/conv_rows.h). */
const char* lcv_conv3d_last_kernel(void);
/* Strided conv3d for the VAE ENCODER's downsampling stages (upstream WanResample "downsample2d/3d": ZeroPad2d((0,1,0,1))
 * + 3x3 stride-2 conv per frame; (3,1,1) stride-2 temporal conv over [cached last frame | chunk]; contract of
 * vae.encode at delta_experiment/scripts/common.py:158-174).  Output pixel (t,h,w) reads input (t*st+dt, h*sh+dh, w*sw+dw),
 * no front padding, taps past the input extent read zeros; the caller gives the output extent. */
int lcv_conv3d_strided(const void* x, const void* w, const void* bias, void* out, const void* zero_page, int64_t B,
                       int64_t Tin, int64_t Hin, int64_t Win, int64_t Cin, int64_t Cout, int64_t ldc, int kt, int kh,
                       int kw, int st, int sh, int sw, int64_t Tout, int64_t Hout, int64_t Wout, void* stream);
/* WAN RMS_norm over channels (channels-last rows padded to Cpad): y = x / max(||x||_2, 1e-12) * sqrt(C) * gamma,
 * optional SiLU; padding channels are written as zeros. */
int lcv_vae_rmsnorm_silu(const void* x, const void* gamma, void* y, int64_t rows, int64_t C, int64_t Cpad,
                         int apply_silu, void* stream);
/* p = softmax(scale * s) row-wise; s fp32 [rows, n] (row stride ld_s), p bf16 [rows, ld_p] (columns >= n zeroed). */
int lcv_softmax_rows(const float* s, void* p, int64_t rows, int64_t n, int64_t ld_s, int64_t ld_p, float scale,
                     void* stream);

/* ---- on-device evaluation of generated frames -------------------------------------------------
 * Replaces the host loops of evaluate_generation_metrics (delta_experiment/scripts/common.py:663-757) and of the
 * baseline runner (baseline_experiment/scripts/run_baseline.py:124-145, 436-441): per-frame sum((gen-gt)^2) for PSNR
 * and a separable-window SSIM map summed over the windows that lie inside the frame — which is what survives
 * torchmetrics' pad-5/crop-5 (`StructuralSimilarityIndexMeasure(data_range=1.0)`: win 11, Gaussian sigma 1.5,
 * cov_norm 1, clamp_var 1; common.py:760-764) and skimage's crop of (win-1)/2 (`structural_similarity` defaults:
 * win 7, uniform taps 1/7, cov_norm 49/48, clamp_var 0; run_baseline.py:135-136).
 * Frames are NHWC [N,H,W,C], C interleaved (C <= 4); `gen` fp32 in [0,1]; `gt` fp32 or raw uint8 (gt_is_u8: divided
 * by 255 in the kernel).  Each call writes fp32 partial sums, [N, n_sqerr] / [N, n_ssim] as sized by
 * lcv_frame_metric_partials; the caller adds them (fp64) and divides by H*W*C, resp. (H-win+1)*(W-win+1)*C. */
int lcv_frame_metric_partials(int64_t H, int64_t W, int64_t C, int win, int64_t* n_sqerr, int64_t* n_ssim);
int lcv_frame_sqerr(const float* gen, const void* gt, int gt_is_u8, float* partials, int64_t N, int64_t E, void* stream);
int lcv_frame_ssim(const float* gen, const void* gt, int gt_is_u8, float* partials, int64_t N, int64_t H, int64_t W,
                   int64_t C, const float* window /* host pointer, `win` normalised taps */, int win, float cov_norm,
                   int clamp_var, float c1, float c2, void* stream);

/* ---- UMT5 text encoder ------------------------------------------------------------------------
 * The non-GEMM pieces of transformers.UMT5EncoderModel, which the reference runs once per prompt
 * (delta_experiment/scripts/common.py:62-64, 228-255); the linears go through lcv_gemm_nt.
 * lcv_gather_rows: out[r, :] = table[ids[r], :] (bf16 rows of C).
 * lcv_t5_rmsnorm: y = bf16(w * bf16(x * rsqrt(mean(x^2) + eps))) — T5LayerNorm: fp32 statistics, no mean, no bias.
 * lcv_geglu_tanh_fwd: out = bf16(bf16(gelu_new(gate)) * up), rows x F, inputs with row stride ld_in.
 * lcv_t5_attention: out[b, i, h*64:(h+1)*64] = softmax_j(q_i.k_j + bias_by_dist[h, j - i + S - 1] + mask_j) . v_j with
 * d_kv = 64 and NO 1/sqrt(d) scaling; q/k/v are bf16 views with row stride ld_qkv and batch stride bs_qkv (e.g. the three
 * column blocks of one fused projection), key_mask int32 [B, S] (1 = attend), S <= 512. */
int lcv_gather_rows(const void* table, const int64_t* ids, void* out, int64_t n, int64_t C, int64_t vocab, void* stream);
int lcv_t5_rmsnorm(const void* x, const void* w, void* y, int64_t rows, int64_t C, float eps, void* stream);
int lcv_geglu_tanh_fwd(const void* gate, const void* up, void* out, int64_t rows, int64_t F, int64_t ld_in, void* stream);
int lcv_t5_attention(const void* q, const void* k, const void* v, void* out, const float* bias_by_dist,
                     const int* key_mask, int64_t B, int64_t S, int64_t H, int64_t ld_qkv, int64_t ld_o, int64_t bs_qkv,
                     int64_t bs_o, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* LCV_HIP_H */
