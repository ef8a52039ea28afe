"""Differentiable front-ends of the HIP ops.

Each op runs the forward kernel directly when no gradient is needed (denoise path) and goes
through a `torch.autograd.Function` whose backward is another HIP kernel when it is (TTA inner
loop).  Frozen base weights never get a gradient (LoRA-only backward): `dx = dy @ W` uses a
resident transposed copy of W so that forward and backward share the one NT GEMM kernel.
"""
import weakref
from typing import List, Optional

import torch

from . import ops
from .lib import (LCV_EPI_GELU_TANH, LCV_EPI_NONE, LCV_EPI_SWIGLU, LcvError)

BF16 = torch.bfloat16
F32 = torch.float32

_EPI = {None: LCV_EPI_NONE, "gelu_tanh": LCV_EPI_GELU_TANH}


def _needs_grad(*ts) -> bool:
    return torch.is_grad_enabled() and any(t is not None and t.requires_grad for t in ts)


# --------------------------------------------------------------------------- transposed weights
_WT_CACHE = {}  # id(weight) -> (weakref to the weight, version, W^T)


def transposed_weight(w: torch.Tensor) -> torch.Tensor:
    """Resident W^T ([in, out], contiguous) for dx = dy @ W.  Entries are tied to the weight OBJECT (a freed model's
    storage address can be reused by the next one) and rebuilt when the weight is modified in place."""
    key = id(w)
    ent = _WT_CACHE.get(key)
    if ent is not None and ent[0]() is w and ent[1] == w._version and ent[2].data_ptr() != 0:
        return ent[2]
    with torch.no_grad():
        wt = w.detach().t().contiguous()
    _WT_CACHE[key] = (weakref.ref(w, lambda _r, k=key: _WT_CACHE.pop(k, None)), w._version, wt)
    return wt


def clear_weight_caches():
    _WT_CACHE.clear()
    _XT_LAST.clear()


# full-model TTA: the token-major input of a trainable linear, transposed for the dense weight gradient.  Linears that
# share their input (w1 / w3) run back to back in the backward, so the last transpose is kept.
_XT_LAST = {}


def _transposed_input(x: torch.Tensor) -> torch.Tensor:
    key = (x.data_ptr(), x._version, tuple(x.shape))
    ent = _XT_LAST.get("x")
    if ent is not None and ent[0] == key:
        return ent[1]
    xt = ops.transpose_pad(x)
    _XT_LAST["x"] = (key, xt)
    return xt


def _dense_param_grads(x2: torch.Tensor, dyb: torch.Tensor, want_dw: bool, want_db: bool, w_dtype, b_dtype):
    """(dW [N,K], db [N]) of y = x W^T + b from x [M,K] and dy [M,N] (run_full_tta.py: every weight is trainable)."""
    dyT = ops.transpose_pad(dyb)
    dw = ops.dense_wgrad(dyT, _transposed_input(x2)).to(w_dtype) if want_dw else None
    db = ops.rowsum(dyT).to(b_dtype) if want_db else None
    return dw, db


# --------------------------------------------------------------------------- linear
def _pad_cols(t: torch.Tensor, width: int = 64) -> torch.Tensor:
    out = torch.zeros((t.shape[0], width), dtype=t.dtype, device=t.device)
    out[:, : t.shape[1]].copy_(t)
    return out


class _LinearFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w, b, epi, out_f32):
        ctx.save_for_backward(x if w.requires_grad else None, w)
        ctx.out_f32 = out_f32
        ctx.b_dtype = b.dtype if b is not None else None
        if epi != LCV_EPI_NONE:
            raise LcvError("linear: fused activation epilogues have no backward (use the unfused form when training)")
        return ops.gemm_nt(x, w, b, epilogue=epi, out_f32=out_f32)

    @staticmethod
    def backward(ctx, dy):
        x, w = ctx.saved_tensors
        dyb = (dy if dy.dtype == BF16 else dy.to(BF16)).contiguous()
        N, K = w.shape
        dx = dw = db = None
        if ctx.needs_input_grad[0]:
            # a frozen base weight keeps a resident W^T (LoRA TTA); a trainable one changes every step, so its
            # transpose is a temporary
            if not w.requires_grad:
                wt = transposed_weight(w)
            elif N % 64 == 0 and w.is_contiguous():
                wt = ops.transpose_pad(w.detach())      # [K, N]: the HBM-bound transpose kernel, no padding needed
            else:
                wt = w.detach().t().contiguous()
            dx = ops.gemm_nt(dyb, wt, None)
        want_db = ctx.needs_input_grad[2]
        if w.requires_grad and N <= 32 and not want_db:          # adapter-sized: the skinny kernels
            dw = ops.tn_skinny(_pad_cols(dyb), x, N).to(w.dtype)
        elif w.requires_grad and K <= 32 and not want_db:
            dw = ops.tn_skinny(_pad_cols(x), dyb, K).t().contiguous().to(w.dtype)
        elif w.requires_grad or want_db:                          # dense weights / biases: full-model TTA
            dw, db = _dense_param_grads(x if w.requires_grad else None, dyb, w.requires_grad, want_db, w.dtype, ctx.b_dtype)
        return dx, dw, db, None, None


def linear(x: torch.Tensor, w: torch.Tensor, b: Optional[torch.Tensor], epilogue: Optional[str] = None,
           out_f32: bool = False) -> torch.Tensor:
    x = x if x.dtype == BF16 else x.to(BF16)
    if not x.is_contiguous():
        x = x.contiguous()
    epi = _EPI[epilogue]
    if _needs_grad(x, w, b):
        return _LinearFn.apply(x, w, b, epi, out_f32)
    return ops.gemm_nt(x, w, b, epilogue=epi, out_f32=out_f32)


class _LinearF32Fn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, a, w, b, act_in):
        ctx.save_for_backward(a, w)
        ctx.act_in = act_in
        ctx.b_dtype = b.dtype if b is not None else None
        return ops.linear_f32_smallm(a, w, b, act_in)

    @staticmethod
    def backward(ctx, dy):
        a, w = ctx.saved_tensors
        dw = db = None
        if w.requires_grad or ctx.needs_input_grad[2]:            # full-model TTA: adaLN modulation / timestep MLP weights
            dw, db = ops.linear_f32_smallm_wgrad(dy.contiguous().float(), a, ctx.act_in, ctx.needs_input_grad[2])
            dw = dw.to(w.dtype) if w.requires_grad else None
            db = db.to(ctx.b_dtype) if db is not None else None
        da = ops.linear_f32_smallm_bwd(dy.contiguous(), w, a, ctx.act_in) if ctx.needs_input_grad[0] else None
        return da, dw, db, None


def linear_f32(a: torch.Tensor, w: torch.Tensor, b: Optional[torch.Tensor], act_in: int = 0) -> torch.Tensor:
    if _needs_grad(a, w, b):
        return _LinearF32Fn.apply(a.contiguous(), w, b, act_in)
    return ops.linear_f32_smallm(a, w, b, act_in)


# --------------------------------------------------------------------------- norms / residual
class _AdaLNFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, mod, shift_idx, scale_idx, T, eps):
        ctx.save_for_backward(x, mod)
        ctx.args = (shift_idx, scale_idx, T, eps)
        return ops.adaln_modulate(x, mod, shift_idx, scale_idx, T, eps)

    @staticmethod
    def backward(ctx, dy):
        x, mod = ctx.saved_tensors
        shift_idx, scale_idx, T, eps = ctx.args
        dx, dmod = ops.adaln_modulate_bwd(x, mod, dy.contiguous(), shift_idx, scale_idx, T, eps,
                                          need_dmod=ctx.needs_input_grad[1])
        return dx, dmod, None, None, None, None


class _AdaLNForkFn(torch.autograd.Function):
    """`(modulate(LN(x)), x)`: the norm at the entry of a residual branch together with the branch's residual input.  The
    second output aliases x; in the backward the gradient that comes down the residual path is added to the norm's own dx
    INSIDE the norm-backward kernel (one fp32 add before the single bf16 rounding) instead of by a separate elementwise
    kernel of autograd's — three such forks per block and step."""

    @staticmethod
    def forward(ctx, x, mod, shift_idx, scale_idx, T, eps):
        ctx.save_for_backward(x, mod)
        ctx.args = (shift_idx, scale_idx, T, eps)
        ctx.set_materialize_grads(False)
        return ops.adaln_modulate(x, mod, shift_idx, scale_idx, T, eps), x.view_as(x)

    @staticmethod
    def backward(ctx, dy, dres):
        x, mod = ctx.saved_tensors
        shift_idx, scale_idx, T, eps = ctx.args
        if dy is None:
            return dres, None, None, None, None, None
        dx, dmod = ops.adaln_modulate_bwd(x, mod, dy.contiguous(), shift_idx, scale_idx, T, eps,
                                          need_dmod=ctx.needs_input_grad[1], dres=dres)
        return dx, dmod, None, None, None, None


def adaln_modulate_fork(x, mod, shift_idx, scale_idx, T, eps=1e-6):
    """-> (x_modulated, x_residual).  Training form of `adaln_modulate` for a residual branch (see _AdaLNForkFn)."""
    if _needs_grad(x, mod):
        return _AdaLNForkFn.apply(x, mod, shift_idx, scale_idx, T, eps)
    return ops.adaln_modulate(x, mod, shift_idx, scale_idx, T, eps), x


def adaln_modulate(x, mod, shift_idx, scale_idx, T, eps=1e-6):
    if _needs_grad(x, mod):
        return _AdaLNFn.apply(x, mod, shift_idx, scale_idx, T, eps)
    return ops.adaln_modulate(x, mod, shift_idx, scale_idx, T, eps)


class _LayerNormAffineFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w, b, eps):
        ctx.save_for_backward(x, w)
        ctx.eps = eps
        return ops.layernorm_affine(x, w, b, eps)

    @staticmethod
    def backward(ctx, dy):
        x, w = ctx.saved_tensors
        dx, dw, db = ops.layernorm_affine_bwd(x, w, dy.contiguous(), ctx.eps,
                                              need_dw=ctx.needs_input_grad[1] or ctx.needs_input_grad[2])
        return dx, (dw.to(w.dtype) if dw is not None else None), (db.to(w.dtype) if db is not None else None), None


class _LayerNormAffineForkFn(torch.autograd.Function):
    """`(LN(x) * w + b, x)` for the cross-attention branch: as _AdaLNForkFn."""

    @staticmethod
    def forward(ctx, x, w, b, eps):
        ctx.save_for_backward(x, w)
        ctx.eps = eps
        ctx.set_materialize_grads(False)
        return ops.layernorm_affine(x, w, b, eps), x.view_as(x)

    @staticmethod
    def backward(ctx, dy, dres):
        x, w = ctx.saved_tensors
        if dy is None:
            return dres, None, None, None
        dx, dw, db = ops.layernorm_affine_bwd(x, w, dy.contiguous(), ctx.eps,
                                              need_dw=ctx.needs_input_grad[1] or ctx.needs_input_grad[2], dres=dres)
        return dx, (dw.to(w.dtype) if dw is not None else None), (db.to(w.dtype) if db is not None else None), None


def layernorm_affine_fork(x, w, b, eps=1e-6):
    if _needs_grad(x, w, b):
        return _LayerNormAffineForkFn.apply(x, w, b, eps)
    return ops.layernorm_affine(x, w, b, eps), x


def layernorm_affine(x, w, b, eps=1e-6):
    if _needs_grad(x, w, b):
        return _LayerNormAffineFn.apply(x, w, b, eps)
    return ops.layernorm_affine(x, w, b, eps)


class _GateResidualFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, y, mod, gate_idx, T):
        ctx.save_for_backward(y, mod)
        ctx.args = (gate_idx, T)
        return ops.gate_residual(x, y, mod, gate_idx, T)

    @staticmethod
    def backward(ctx, dout):
        y, mod = ctx.saved_tensors
        gate_idx, T = ctx.args
        dout = dout.contiguous()
        if mod is None:
            return dout, dout, None, None, None
        dy, dmod = ops.gate_residual_bwd(y, mod, dout, gate_idx, T, need_dmod=ctx.needs_input_grad[2])
        return dout, dy, dmod, None, None


def gate_residual(x, y, mod, gate_idx, T):
    if _needs_grad(x, y, mod):
        return _GateResidualFn.apply(x, y, mod, gate_idx, T)
    return ops.gate_residual(x, y, mod, gate_idx, T)


class _PadFrontZeroFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, n):
        ctx.n = n
        B, N, C = x.shape
        out = x.new_zeros((B, N + n, C))
        out[:, n:].copy_(x)
        return out

    @staticmethod
    def backward(ctx, dout):
        return dout[:, ctx.n:].contiguous(), None


def pad_front_zero(x: torch.Tensor, n: int) -> torch.Tensor:
    """[B, N, C] -> [B, n + N, C] with n leading zero rows (memory movement only)."""
    if _needs_grad(x):
        return _PadFrontZeroFn.apply(x, n)
    B, N, C = x.shape
    out = x.new_zeros((B, N + n, C))
    out[:, n:].copy_(x)
    return out


# --------------------------------------------------------------------------- attention
def _attend_regions(q, k, v, scale, n_cond, need_lse):
    """q,k,v [B,N,H,D] views; conditioning tokens attend conditioning tokens only."""
    B, N, H, D = q.shape
    o = torch.empty((B, N, H, D), dtype=BF16, device=q.device)
    lses = []
    if n_cond > 0:
        _, l1 = ops.attention(q[:, :n_cond], k[:, :n_cond], v[:, :n_cond], scale, out=o[:, :n_cond], need_lse=need_lse)
        lses.append(l1)
        if N > n_cond:
            _, l2 = ops.attention(q[:, n_cond:], k, v, scale, out=o[:, n_cond:], need_lse=need_lse)
            lses.append(l2)
    else:
        _, l1 = ops.attention(q, k, v, scale, out=o, need_lse=need_lse)
        lses.append(l1)
    return o, lses


class _SelfAttentionFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, qkv, wq, wk, cs, scale, n_cond, eps):
        B, N, _, H, D = qkv.shape
        qk = torch.empty((B, N, 2, H, D), dtype=BF16, device=qkv.device)  # roped q, k (qkv kept for the norm backward)
        # q is produced pre-scaled into log2 units (ops.log2_qscale) and the attention kernels run with scale = ln 2
        ops.qknorm_rope(qkv[:, :, 0], qkv[:, :, 1], None, qk[:, :, 0], qk[:, :, 1], None, wq, wk, cs, 0, eps,
                        q_scale=ops.log2_qscale(scale))
        o, lses = _attend_regions(qk[:, :, 0], qk[:, :, 1], qkv[:, :, 2], ops.LN2, n_cond, True)
        ctx.save_for_backward(qkv, qk, o, wq, wk, cs, *lses)
        ctx.args = (scale, n_cond, eps)
        return o

    @staticmethod
    def backward(ctx, do):
        qkv, qk, o, wq, wk, cs, *lses = ctx.saved_tensors
        scale, n_cond, eps = ctx.args
        want_dw = wq.requires_grad or wk.requires_grad   # norm-weight tuning (run_norm_tune_tta.py, --norm-target qk_norm)
        B, N, _, H, D = qkv.shape
        do = do.contiguous()
        # dV needs no further transform: the kernels write it straight into the gradient of the packed qkv (strided view);
        # dQ / dK (w.r.t. the roped, normalised q and k) go through the norm / RoPE backward below.  No zero-fill: the pass that
        # sees EVERY key (noise queries) runs first and overwrites, the cond x cond pass then adds into the cond rows.
        dqkv = torch.empty_like(qkv)
        dqk_r = torch.empty((B, N, 2, H, D), dtype=BF16, device=qkv.device)
        q, k, v = qk[:, :, 0], qk[:, :, 1], qkv[:, :, 2]
        dq, dk, dv = dqk_r[:, :, 0], dqk_r[:, :, 1], dqkv[:, :, 2]
        if n_cond > 0 and N > n_cond:
            ops.attention_bwd(q[:, n_cond:], k, v, o[:, n_cond:], do[:, n_cond:], lses[1],
                              dq[:, n_cond:], dk, dv, ops.LN2, accumulate_kv=False)
            ops.attention_bwd(q[:, :n_cond], k[:, :n_cond], v[:, :n_cond], o[:, :n_cond], do[:, :n_cond], lses[0],
                              dq[:, :n_cond], dk[:, :n_cond], dv[:, :n_cond], ops.LN2, accumulate_kv=True)
        else:   # no conditioning tokens, or nothing but conditioning tokens: one region
            ops.attention_bwd(q, k, v, o, do, lses[0], dq, dk, dv, ops.LN2, accumulate_kv=False)
        dwq = torch.zeros(D, dtype=torch.float32, device=qkv.device) if want_dw else None
        dwk = torch.zeros(D, dtype=torch.float32, device=qkv.device) if want_dw else None
        ops.qknorm_rope_bwd(qkv[:, :, 0], qkv[:, :, 1], dq, dk, dqkv[:, :, 0], dqkv[:, :, 1], wq, wk, cs, 0, eps,
                            q_scale=ops.log2_qscale(scale), dwq=dwq, dwk=dwk)
        return (dqkv, dwq.to(wq.dtype) if (want_dw and wq.requires_grad) else None,
                dwk.to(wk.dtype) if (want_dw and wk.requires_grad) else None, None, None, None, None)


def self_attention(qkv, wq, wk, cs, scale, n_cond, eps, return_kv=False):
    """qkv [B,N,3,H,D] (fresh GEMM output).  Returns (o [B,N,H,D], kv) where kv = (K roped, V) of the
    sequence as [B,N,H,D] copies when requested (conditioning-token KV cache)."""
    if _needs_grad(qkv, wq, wk):
        if return_kv:
            raise LcvError("self_attention: return_kv is an inference-only path")
        return _SelfAttentionFn.apply(qkv, wq, wk, cs, scale, n_cond, eps), None
    q, k, v = qkv[:, :, 0], qkv[:, :, 1], qkv[:, :, 2]
    ops.qknorm_rope(q, k, None, q, k, None, wq, wk, cs, 0, eps, q_scale=ops.log2_qscale(scale))  # in place on the GEMM output
    o, _ = _attend_regions(q, k, v, ops.LN2, n_cond, False)
    kv = (k.contiguous(), v.contiguous()) if return_kv else None
    return o, kv


class _SPSelfAttentionFn(torch.autograd.Function):
    """Frame-sharded self-attention with its adjoint: forward all-gathers the roped K and V and runs
    local-Q x full-KV; backward runs the two-pass kernels on (local Q, full K/V), sums dK / dV over the ranks and keeps this
    rank's rows (the adjoint of the all-gather), then the q/k-norm + RoPE backward on the local rows.  Conditioning tokens
    (the first `n_cond_glob` keys; `n_cond_loc` of them are local queries, always a prefix) attend conditioning keys only."""

    @staticmethod
    def forward(ctx, qkv, wq, wk, cs, scale, eps, sp, n_cond_loc, n_cond_glob):
        B, N, _, H, D = qkv.shape
        qk = torch.empty((B, N, 2, H, D), dtype=BF16, device=qkv.device)
        ops.qknorm_rope(qkv[:, :, 0], qkv[:, :, 1], None, qk[:, :, 0], qk[:, :, 1], None, wq, wk, cs, sp.token_offset, eps,
                        q_scale=ops.log2_qscale(scale))
        k_full, v_full = sp.all_gather_kv(qk[:, :, 1].contiguous(), qkv[:, :, 2].contiguous())
        q = qk[:, :, 0]
        o = torch.empty((B, N, H, D), dtype=BF16, device=qkv.device)
        lses = []
        if n_cond_loc > 0:
            _, l1 = ops.attention(q[:, :n_cond_loc], k_full[:, :n_cond_glob], v_full[:, :n_cond_glob], ops.LN2,
                                  out=o[:, :n_cond_loc], need_lse=True)
            lses.append(l1)
        if N > n_cond_loc:
            _, l2 = ops.attention(q[:, n_cond_loc:], k_full, v_full, ops.LN2, out=o[:, n_cond_loc:], need_lse=True)
            lses.append(l2)
        ctx.save_for_backward(qkv, qk, k_full, v_full, o, wq, wk, cs, *lses)
        ctx.args = (scale, eps, sp, n_cond_loc, n_cond_glob)
        return o

    @staticmethod
    def backward(ctx, do):
        qkv, qk, k_full, v_full, o, wq, wk, cs, *lses = ctx.saved_tensors
        scale, eps, sp, n_cond_loc, n_cond_glob = ctx.args
        B, N, _, H, D = qkv.shape
        do = do.contiguous()
        q = qk[:, :, 0]
        dq = torch.zeros((B, N, H, D), dtype=BF16, device=qkv.device)
        dk_full = sp.padded_zeros(k_full)       # allocated in the reduce-scatter's layout (pads at the end)
        dv_full = sp.padded_zeros(v_full)
        i = 0
        if n_cond_loc > 0:
            ops.attention_bwd(q[:, :n_cond_loc], k_full[:, :n_cond_glob], v_full[:, :n_cond_glob], o[:, :n_cond_loc],
                              do[:, :n_cond_loc], lses[i], dq[:, :n_cond_loc], dk_full[:, :n_cond_glob],
                              dv_full[:, :n_cond_glob], ops.LN2, accumulate_kv=False)
            i += 1
        if N > n_cond_loc:
            ops.attention_bwd(q[:, n_cond_loc:], k_full, v_full, o[:, n_cond_loc:], do[:, n_cond_loc:], lses[i],
                              dq[:, n_cond_loc:], dk_full, dv_full, ops.LN2, accumulate_kv=(n_cond_loc > 0))
        dk = sp.reduce_scatter_kv(dk_full)
        dv = sp.reduce_scatter_kv(dv_full)
        dqkv = torch.empty_like(qkv)
        want_dw = wq.requires_grad or wk.requires_grad
        dwq = torch.zeros(D, dtype=torch.float32, device=qkv.device) if want_dw else None
        dwk = torch.zeros(D, dtype=torch.float32, device=qkv.device) if want_dw else None
        ops.qknorm_rope_bwd(qkv[:, :, 0], qkv[:, :, 1], dq, dk, dqkv[:, :, 0], dqkv[:, :, 1], wq, wk, cs, sp.token_offset, eps,
                            q_scale=ops.log2_qscale(scale), dwq=dwq, dwk=dwk)
        dqkv[:, :, 2].copy_(dv)
        return (dqkv, dwq.to(wq.dtype) if (want_dw and wq.requires_grad) else None,
                dwk.to(wk.dtype) if (want_dw and wk.requires_grad) else None, None, None, None, None, None, None)


def sp_self_attention(qkv, wq, wk, cs, scale, eps, sp, n_cond_loc, n_cond_glob):
    """Sequence-parallel self-attention on a fresh local qkv [B, n_local, 3, H, D]; autograd-aware."""
    if _needs_grad(qkv, wq, wk):
        return _SPSelfAttentionFn.apply(qkv, wq, wk, cs, scale, eps, sp, n_cond_loc, n_cond_glob)
    B, N, _, H, D = qkv.shape
    q, k, v = qkv[:, :, 0], qkv[:, :, 1], qkv[:, :, 2]
    ops.qknorm_rope(q, k, None, q, k, None, wq, wk, cs, sp.token_offset, eps, q_scale=ops.log2_qscale(scale))
    if n_cond_glob == 0 and sp.overlap and qkv.is_cuda and sp.world > 1:
        return _sp_attention_overlapped(q, k.contiguous(), v.contiguous(), sp)
    k_full, v_full = sp.all_gather_kv(k.contiguous(), v.contiguous())
    if n_cond_glob == 0:
        o, _ = ops.attention(q, k_full, v_full, ops.LN2)
        return o
    o = torch.empty((B, N, H, D), dtype=BF16, device=qkv.device)
    if n_cond_loc > 0:
        ops.attention(q[:, :n_cond_loc], k_full[:, :n_cond_glob], v_full[:, :n_cond_glob], ops.LN2, out=o[:, :n_cond_loc])
    if N > n_cond_loc:
        ops.attention(q[:, n_cond_loc:], k_full, v_full, ops.LN2, out=o[:, n_cond_loc:])
    return o


def _sp_attention_overlapped(q, k_loc, v_loc, sp):
    """LCV_SP_OVERLAP=1 (inference, no conditioning split): the K/V all-gather runs on a side stream while the flash kernel
    attends the LOCAL keys; the keys of the ranks before and after follow once the gather has landed, and the partial
    results are merged through their log-sum-exps:  o = sum_i o_i * exp(lse_i - lse),  lse = logsumexp_i(lse_i).
    Same softmax, partitioned over key ranges: equal to the one-call form up to rounding, NOT bit-identical."""
    main = torch.cuda.current_stream()
    side = sp.side_stream(q.device)
    side.wait_stream(main)                                   # K / V of this layer are final on the main stream
    with torch.cuda.stream(side):
        k_full, v_full = sp.all_gather_kv(k_loc, v_loc)
    parts = [ops.attention(q, k_loc, v_loc, ops.LN2, need_lse=True)]          # overlaps the collective
    main.wait_stream(side)
    k_full.record_stream(main); v_full.record_stream(main)
    lo, hi = sp.token_offset, sp.token_offset + k_loc.shape[1]
    if lo > 0:
        parts.append(ops.attention(q, k_full[:, :lo], v_full[:, :lo], ops.LN2, need_lse=True))
    if hi < k_full.shape[1]:
        parts.append(ops.attention(q, k_full[:, hi:], v_full[:, hi:], ops.LN2, need_lse=True))
    return merge_attention_parts(parts)


def merge_attention_parts(parts):
    """[(o_i bf16 [B, N, H, D], lse_i fp32 [B, H, N])] of ONE query set over disjoint key ranges -> the attention over their
    union: o = sum_i o_i * exp(lse_i - lse), lse = logsumexp_i(lse_i).  `lse_i` is what `ops.attention(..., need_lse=True)`
    returns: the NATURAL logarithm of the row's sum of exp(scale * q.k) (the kernels keep m_run * scale + ln(l), also when
    they run in log2 units internally with scale = ln 2) - pinned by tests/test_gpu_kernels.py on one GPU."""
    if len(parts) == 1:
        return parts[0][0]
    lses = torch.stack([l for _, l in parts], 0)                               # [P, B, H, N] natural-log units
    lse = torch.logsumexp(lses, dim=0)
    out = None
    for (o_i, l_i) in parts:
        w_i = torch.exp(l_i - lse).permute(0, 2, 1).unsqueeze(-1)              # [B, N, H, 1]
        out = o_i.float() * w_i if out is None else out + o_i.float() * w_i
    return out.to(BF16)


def cached_attention(qkv, k_c, v_c, wq, wk, cs, scale, eps):
    """Noise-token queries against [cached cond K/V | fresh K/V]; inference only."""
    if _needs_grad(qkv):
        raise LcvError("cached_attention: the KV-cached step is an inference-only path")
    B, N, _, H, D = qkv.shape
    n_c = k_c.shape[1]
    kbuf = torch.empty((B, n_c + N, H, D), dtype=BF16, device=qkv.device)
    vbuf = torch.empty((B, n_c + N, H, D), dtype=BF16, device=qkv.device)
    kbuf[:, :n_c].copy_(k_c if k_c.shape[0] == B else k_c.expand(B, -1, -1, -1))
    vbuf[:, :n_c].copy_(v_c if v_c.shape[0] == B else v_c.expand(B, -1, -1, -1))
    q = qkv[:, :, 0]
    ops.qknorm_rope(q, qkv[:, :, 1], qkv[:, :, 2], q, kbuf[:, n_c:], vbuf[:, n_c:], wq, wk, cs, n_c, eps,
                    q_scale=ops.log2_qscale(scale))
    o, _ = ops.attention(q, kbuf, vbuf, ops.LN2)
    return o


class _CrossAttentionFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, q_raw, kv_raw, wq, wk, seqlens, scale, eps):
        B, N, H, D = q_raw.shape
        L = kv_raw.shape[1]
        qn = torch.empty_like(q_raw)
        kn = torch.empty((1, L, H, D), dtype=BF16, device=q_raw.device)
        ops.qknorm_rope(q_raw, None, None, qn, None, None, wq, wk, None, 0, eps)
        ops.qknorm_rope(None, kv_raw[:, :, 0], None, None, kn, None, wq, wk, None, 0, eps)
        o = torch.empty_like(q_raw)
        lses, off = [], 0
        for b in range(B):
            Lb = int(seqlens[b])
            _, l = ops.attention(qn[b:b + 1], kn[:, off:off + Lb], kv_raw[:, off:off + Lb, 1], scale, out=o[b:b + 1],
                                 need_lse=True)
            lses.append(l)
            off += Lb
        ctx.save_for_backward(q_raw, kv_raw, qn, kn, o, wq, wk, *lses)
        ctx.args = (list(seqlens), scale, eps)
        return o

    @staticmethod
    def backward(ctx, do):
        q_raw, kv_raw, qn, kn, o, wq, wk, *lses = ctx.saved_tensors
        seqlens, scale, eps = ctx.args
        want_dw = wq.requires_grad or wk.requires_grad
        B, N, H, D = q_raw.shape
        L = kv_raw.shape[1]
        do = do.contiguous()
        dqn = torch.empty_like(qn)
        dkv = torch.zeros((1, L, 2, H, D), dtype=BF16, device=q_raw.device)  # [dk (normed) | dv]
        off = 0
        for b in range(B):
            Lb = int(seqlens[b])
            ops.attention_bwd(qn[b:b + 1], kn[:, off:off + Lb], kv_raw[:, off:off + Lb, 1], o[b:b + 1], do[b:b + 1],
                              lses[b], dqn[b:b + 1], dkv[:, off:off + Lb, 0], dkv[:, off:off + Lb, 1], scale,
                              accumulate_kv=False)
            off += Lb
        dq_raw = torch.empty_like(q_raw)
        dwq = torch.zeros(D, dtype=torch.float32, device=q_raw.device) if want_dw else None
        dwk = torch.zeros(D, dtype=torch.float32, device=q_raw.device) if want_dw else None
        ops.qknorm_rope_bwd(q_raw, None, dqn, None, dq_raw, None, wq, wk, None, 0, eps, dwq=dwq)
        dkv_raw = torch.empty_like(kv_raw)
        ops.qknorm_rope_bwd(None, kv_raw[:, :, 0], None, dkv[:, :, 0], None, dkv_raw[:, :, 0], wq, wk, None, 0, eps, dwk=dwk)
        dkv_raw[:, :, 1].copy_(dkv[:, :, 1])
        return (dq_raw, dkv_raw, dwq.to(wq.dtype) if (want_dw and wq.requires_grad) else None,
                dwk.to(wk.dtype) if (want_dw and wk.requires_grad) else None, None, None, None)


def cross_attention(q, kv, wq, wk, seqlens: List[int], scale, eps):
    """q [B,N,H,D] (fresh), kv [1, sum(seqlens), 2, H, D] (fresh) -> o [B,N,H,D]."""
    if _needs_grad(q, kv, wq, wk):
        return _CrossAttentionFn.apply(q, kv, wq, wk, seqlens, scale, eps)
    B, N, H, D = q.shape
    k, v = kv[:, :, 0], kv[:, :, 1]
    ops.qknorm_rope(q, None, None, q, None, None, wq, wk, None, 0, eps)
    ops.qknorm_rope(None, k, None, None, k, None, wq, wk, None, 0, eps)
    o = torch.empty_like(q)
    off = 0
    for b in range(B):
        Lb = int(seqlens[b])
        ops.attention(q[b:b + 1], k[:, off:off + Lb], v[:, off:off + Lb], scale, out=o[b:b + 1])
        off += Lb
    return o


# --------------------------------------------------------------------------- GELU (caption embedder, trainable form)
class _GeluTanhFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        ctx.save_for_backward(x)
        return ops.gelu_tanh(x)

    @staticmethod
    def backward(ctx, dy):
        (x,) = ctx.saved_tensors
        return ops.gelu_tanh_bwd(x, (dy if dy.dtype == BF16 else dy.to(BF16)).contiguous())


def gelu_tanh(x: torch.Tensor) -> torch.Tensor:
    if _needs_grad(x):
        return _GeluTanhFn.apply(x)
    return ops.gelu_tanh(x)


# --------------------------------------------------------------------------- SwiGLU
class _SwiGLUFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, g, u):
        ctx.save_for_backward(g, u)
        return ops.swiglu(g, u)

    @staticmethod
    def backward(ctx, dout):
        g, u = ctx.saved_tensors
        dg, du = ops.swiglu_bwd(g, u, dout.contiguous())
        return dg, du


def swiglu(g, u):
    if _needs_grad(g, u):
        return _SwiGLUFn.apply(g, u)
    return ops.swiglu(g, u)


class _SwiGLUFusedFn(torch.autograd.Function):
    """silu(x w1^T) * (x w3^T) through the interleaved weight with a backward (frozen FFN under LoRA / delta TTA): the GEMM's
    epilogue also leaves the pre-activation (gate | up) rows (what the unfused form writes as two tensors anyway), the
    backward turns dh into d(gate | up) in that layout and ONE GEMM against the resident transposed interleaved weight gives
    dx - instead of w1, w3, swiglu_fwd forward and two GEMMs plus an add backward."""

    @staticmethod
    def forward(ctx, x2, w13):
        gu = torch.empty((x2.shape[0], w13.shape[0]), dtype=BF16, device=x2.device)
        h = ops.gemm_nt(x2, w13, None, epilogue=LCV_EPI_SWIGLU, swiglu_aux=gu)
        ctx.save_for_backward(gu, w13)
        return h

    @staticmethod
    def backward(ctx, dh):
        gu, w13 = ctx.saved_tensors
        dgu = ops.swiglu_bwd_interleaved(gu, dh if dh.is_contiguous() else dh.contiguous())
        return ops.gemm_nt(dgu, transposed_weight(w13), None), None


def swiglu_fused(x2: torch.Tensor, w13: torch.Tensor) -> torch.Tensor:
    """silu(x w1^T) * (x w3^T) in one GEMM (interleaved weight, epilogue in registers).  `w13` is a frozen copy: with a
    gradient wanted for x the differentiable form runs; a trainable w1 / w3 never comes here (layers.FeedForwardSwiGLU)."""
    x2 = x2 if x2.is_contiguous() else x2.contiguous()
    if torch.is_grad_enabled() and x2.requires_grad:
        return _SwiGLUFusedFn.apply(x2, w13)
    return ops.gemm_nt(x2, w13, None, epilogue=LCV_EPI_SWIGLU)


# --------------------------------------------------------------------------- patch embed / unpatchify
class _PatchEmbedFn(torch.autograd.Function):
    """Trainable x_embedder (full-model TTA): the latent input never needs a gradient, the conv weight / bias do."""

    @staticmethod
    def forward(ctx, x, w, b):
        B, C = x.shape[0], w.shape[0]
        kin = w[0].numel()
        kpad = ((kin + 63) // 64) * 64
        w2 = w.reshape(C, kin)
        if kpad != kin:
            w2 = torch.nn.functional.pad(w2, (0, kpad - kin))
        tok = ops.patchify(x if x.dtype == BF16 else x.to(BF16), kpad).view(-1, kpad)
        ctx.save_for_backward(tok)
        ctx.meta = (tuple(w.shape), kin, w.dtype, b.dtype if b is not None else None, w.requires_grad)
        return ops.gemm_nt(tok, w2.contiguous(), b).view(B, -1, C)

    @staticmethod
    def backward(ctx, dy):
        (tok,) = ctx.saved_tensors
        wshape, kin, wdt, bdt, wreq = ctx.meta
        dyb = dy.reshape(-1, dy.shape[-1])
        dyb = (dyb if dyb.dtype == BF16 else dyb.to(BF16)).contiguous()
        dw2, db = _dense_param_grads(tok, dyb, wreq, ctx.needs_input_grad[2], wdt, bdt)
        dw = dw2[:, :kin].reshape(wshape).contiguous() if dw2 is not None else None
        return None, dw, db


def patch_embed(x: torch.Tensor, w: torch.Tensor, b: torch.Tensor) -> torch.Tensor:
    """x [B,Cin,T,H,W] -> [B, N, C]: patchify (k = c*4 + ph*2 + pw) + GEMM with the flattened conv weight."""
    if _needs_grad(x):
        raise LcvError("patch_embed: gradients w.r.t. the latent input are out of scope")
    if _needs_grad(w, b):
        return _PatchEmbedFn.apply(x, w, b)
    B = x.shape[0]
    C = w.shape[0]
    kin = w[0].numel()
    kpad = ((kin + 63) // 64) * 64
    w2 = w.reshape(C, kin)
    if kpad != kin:
        w2 = torch.nn.functional.pad(w2, (0, kpad - kin))
    tok = ops.patchify(x if x.dtype == BF16 else x.to(BF16), kpad)
    y = ops.gemm_nt(tok.view(-1, kpad), w2.contiguous(), b)
    return y.view(B, -1, C)


class _UnpatchifyFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, tok, Cout, T, H, W):
        ctx.args = (Cout, T, H, W, tok.dtype)
        return ops.unpatchify(tok, Cout, T, H, W)

    @staticmethod
    def backward(ctx, dout):
        Cout, T, H, W, dt = ctx.args
        return ops.unpatchify_bwd(dout.contiguous(), Cout, T, H, W).to(dt), None, None, None, None


def unpatchify(tok, Cout, T, H, W):
    if _needs_grad(tok):
        return _UnpatchifyFn.apply(tok, Cout, T, H, W)
    return ops.unpatchify(tok, Cout, T, H, W)
