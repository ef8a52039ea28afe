"""Tensor-level wrappers over the C ABI (forward kernels).

PyTorch is plumbing here: it owns device memory and the stream; every op below
hands raw pointers + sizes to liblcv_hip.so.  Nothing in this module computes
with torch operators.
"""
from typing import Optional, Tuple

import torch

from . import lib as _lib
from .lib import (LCV_EPI_GATE_RESIDUAL, LCV_EPI_GELU_TANH, LCV_EPI_NONE, LCV_EPI_SILU, LCV_EPI_SWIGLU,
                  call)

BF16 = torch.bfloat16
F32 = torch.float32
PROFILE = None  # set to a list by bench.py to collect (start, end, flops, Nq, Nk) per attention launch
PROFILE_BWD = None  # likewise for attention_bwd: (start, end, algorithmic flops = 10 B H Nq Nk D, Nq, Nk) per call


def _ptr(t: Optional[torch.Tensor]):
    return None if t is None else t.data_ptr()


def _stream():
    return torch.cuda.current_stream().cuda_stream


def _req(t: torch.Tensor, dtype, name: str):
    if not t.is_cuda:
        raise _lib.LcvError(f"{name}: tensor must live on the GPU (no CPU path exists)")
    if t.dtype != dtype:
        raise _lib.LcvError(f"{name}: expected {dtype}, got {t.dtype}")


# ----------------------------------------------------------------- norms ---
def adaln_modulate(x: torch.Tensor, mod: torch.Tensor, shift_idx: int, scale_idx: int, T: int,
                   eps: float = 1e-6) -> torch.Tensor:
    """x [B, T*S, C] bf16; mod [B, T, k*C] fp32; chunk indices select shift/scale."""
    _req(x, BF16, "adaln_modulate.x"); _req(mod, F32, "adaln_modulate.mod")
    B, N, C = x.shape
    x = x.contiguous(); mod = mod.contiguous()
    y = torch.empty_like(x)
    call("lcv_adaln_modulate_fwd", _ptr(x), _ptr(mod), _ptr(y), B, T, N // T, C, mod.shape[-1],
         shift_idx * C, scale_idx * C, eps, _stream())
    return y


def layernorm_affine(x: torch.Tensor, w: torch.Tensor, b: torch.Tensor, eps: float = 1e-6) -> torch.Tensor:
    _req(x, BF16, "layernorm_affine.x")
    C = x.shape[-1]
    x = x.contiguous()
    wf = w.detach().to(F32).contiguous(); bf = b.detach().to(F32).contiguous()
    y = torch.empty_like(x)
    call("lcv_layernorm_affine_fwd", _ptr(x), _ptr(wf), _ptr(bf), _ptr(y), x.numel() // C, C, eps, _stream())
    return y


def gate_residual(x: torch.Tensor, y: torch.Tensor, mod: Optional[torch.Tensor], gate_idx: int, T: int) -> torch.Tensor:
    _req(x, BF16, "gate_residual.x"); _req(y, BF16, "gate_residual.y")
    B, N, C = x.shape
    x = x.contiguous(); y = y.contiguous()
    out = torch.empty_like(x)
    if mod is not None:
        _req(mod, F32, "gate_residual.mod")
        mod = mod.contiguous()
        call("lcv_gate_residual_fwd", _ptr(x), _ptr(y), _ptr(mod), _ptr(out), B, T, N // T, C, mod.shape[-1],
             gate_idx * C, _stream())
    else:
        call("lcv_gate_residual_fwd", _ptr(x), _ptr(y), None, _ptr(out), B, 1, N, C, 0, 0, _stream())
    return out


def timestep_embedding(t: torch.Tensor, dim: int, max_period: float = 10000.0) -> torch.Tensor:
    """t fp32 [n] -> [n, dim] fp32 (cos | sin)."""
    _req(t, F32, "timestep_embedding.t")
    t = t.contiguous().view(-1)
    out = torch.empty((t.numel(), dim), dtype=F32, device=t.device)
    call("lcv_timestep_embedding", _ptr(t), _ptr(out), t.numel(), dim, float(max_period), _stream())
    return out


# ------------------------------------------------------- q/k norm + rope ---
def qknorm_rope(q_in: Optional[torch.Tensor], k_in: Optional[torch.Tensor], v_in: Optional[torch.Tensor],
                q_out: Optional[torch.Tensor], k_out: Optional[torch.Tensor], v_out: Optional[torch.Tensor],
                wq: torch.Tensor, wk: torch.Tensor, cs: Optional[torch.Tensor], pos_off: int = 0,
                eps: float = 1e-6, q_scale: float = 1.0) -> None:
    """All tensors are [B, N, H, 128] views with contiguous (H, D); in-place allowed (out is in).
    q_scale multiplies the q output before its bf16 rounding (see `LOG2_QSCALE`)."""
    ref = q_in if q_in is not None else k_in
    B, N, H, D = ref.shape
    if D != 128:
        raise _lib.LcvError("qknorm_rope: head_dim must be 128")

    def chk(t, name):
        if t is None:
            return
        _req(t, BF16, name)
        if t.stride(3) != 1 or t.stride(2) != D:
            raise _lib.LcvError(f"{name}: (H, D) must be contiguous")

    for t, nme in ((q_in, "q_in"), (k_in, "k_in"), (v_in, "v_in"), (q_out, "q_out"), (k_out, "k_out"), (v_out, "v_out")):
        chk(t, "qknorm_rope." + nme)
    ins = [t for t in (q_in, k_in, v_in) if t is not None]
    if any(t.stride(0) != ins[0].stride(0) or t.stride(1) != ins[0].stride(1) for t in ins):
        raise _lib.LcvError("qknorm_rope: q_in/k_in/v_in must share strides")
    kvs = [t for t in (k_out, v_out) if t is not None]
    if kvs and any(t.stride(0) != kvs[0].stride(0) or t.stride(1) != kvs[0].stride(1) for t in kvs):
        raise _lib.LcvError("qknorm_rope: k_out/v_out must share strides")
    if cs is not None:
        _req(cs, F32, "qknorm_rope.cs")
        if cs.shape[0] < pos_off + N:
            raise _lib.LcvError("qknorm_rope: cos/sin table shorter than pos_off + N")
    qo = q_out if q_out is not None else ref
    ko = kvs[0] if kvs else ref
    call("lcv_qknorm_rope_fwd", _ptr(q_in), _ptr(k_in), _ptr(v_in), _ptr(q_out), _ptr(k_out), _ptr(v_out),
         _ptr(wq), _ptr(wk), _ptr(cs), B, N, H, ins[0].stride(0), ins[0].stride(1), qo.stride(0), qo.stride(1),
         ko.stride(0), ko.stride(1), pos_off, eps, q_scale, _stream())


# The self-attention path folds its softmax scale into q: q' = q * scale * log2(e) (in the q norm/RoPE kernel, before
# the bf16 rounding) and calls the attention kernels with scale = ln 2, so that exp(scale * q'.k) == exp2(q'.k).
LOG2E = 1.4426950408889634
LN2 = 0.6931471805599453


def log2_qscale(scale: float) -> float:
    return scale * LOG2E


# -------------------------------------------------------------- attention ---
def attention(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, scale: float,
              out: Optional[torch.Tensor] = None, need_lse: bool = False) -> Tuple[torch.Tensor, Optional[torch.Tensor]]:
    """q [B,Nq,H,128], k/v [B,Nk,H,128] (any strides with contiguous D) -> o [B,Nq,H,128], lse [B,H,Nq]."""
    for t, nme in ((q, "q"), (k, "k"), (v, "v")):
        _req(t, BF16, "attention." + nme)
        if t.shape[-1] != 128 or t.stride(-1) != 1:
            raise _lib.LcvError("attention: head_dim must be 128 and contiguous")
    B, Nq, H, D = q.shape
    Nk = k.shape[1]
    if out is None:
        out = torch.empty((B, Nq, H, D), dtype=BF16, device=q.device)
    lse = torch.empty((B, H, Nq), dtype=F32, device=q.device) if need_lse else None
    if PROFILE is not None:  # bench.py: HIP events on the launch stream around the dominant kernel
        ev0 = torch.cuda.Event(enable_timing=True); ev1 = torch.cuda.Event(enable_timing=True)
        ev0.record()
    call("lcv_attn_fwd", _ptr(q), _ptr(k), _ptr(v), _ptr(out), _ptr(lse), B, H, Nq, Nk,
         q.stride(0), q.stride(1), q.stride(2), k.stride(0), k.stride(1), k.stride(2),
         v.stride(0), v.stride(1), v.stride(2), out.stride(0), out.stride(1), out.stride(2),
         float(scale), _stream())
    if PROFILE is not None:
        ev1.record()
        PROFILE.append((ev0, ev1, 4.0 * B * H * Nq * Nk * D, Nq, Nk, _lib.load().lcv_attn_fwd_last_kernel().decode()))
    return out, lse


# ------------------------------------------------------------------ GEMMs ---
_GEMM_WS = {}   # device index -> the split-K tail workspace handed to the library (kept alive here)
GEMM_WS_BYTES = 256 << 20


def _ensure_gemm_workspace(device: torch.device) -> None:
    """Give liblcv_hip.so its split-K tail workspace once per process (one process per GPU): the library never allocates."""
    idx = device.index if device.index is not None else torch.cuda.current_device()
    if idx not in _GEMM_WS:
        if _GEMM_WS:                       # a second device in the same process: the library holds ONE workspace
            raise _lib.LcvError(
                f"gemm_nt on cuda:{idx}, but this process already bound the library's split-K workspace to cuda:"
                f"{next(iter(_GEMM_WS))}: the build runs one process per GPU (torch.distributed), a second device in the "
                "same process would write its partial sums into the first device's memory")
        ws = torch.empty(GEMM_WS_BYTES, dtype=torch.uint8, device=device)
        call("lcv_gemm_set_workspace", _ptr(ws), GEMM_WS_BYTES)
        _GEMM_WS[idx] = ws


# Bumped by every fused optimizer step: those kernels write parameters through raw pointers, which never touches a tensor's
# `_version`.  Whoever caches something derived from TRAINABLE weights (the interleaved SwiGLU weight) keys it on this.
PARAM_EPOCH = 0


def gemm_nt(a: torch.Tensor, w: torch.Tensor, bias: Optional[torch.Tensor] = None, *,
            a2: Optional[torch.Tensor] = None, w2: Optional[torch.Tensor] = None,
            epilogue: int = LCV_EPI_NONE, out_f32: bool = False, resid: Optional[torch.Tensor] = None,
            mod: Optional[torch.Tensor] = None, gate_idx: int = 0, rows_per_frame: int = 1,
            out: Optional[torch.Tensor] = None, swiglu_aux: Optional[torch.Tensor] = None) -> torch.Tensor:
    """c[M,N] = a[M,K] @ w[N,K]^T (+ a2 @ w2^T) + bias with a fused epilogue.  `swiglu_aux` (SwiGLU epilogue only): a
    contiguous bf16 [M, N] tensor that receives the pre-activation (gate | up) rows for the backward."""
    _req(a, BF16, "gemm_nt.a"); _req(w, BF16, "gemm_nt.w")
    if a.shape[0] >= 2048:
        _ensure_gemm_workspace(a.device)
    if a.dim() != 2 or w.dim() != 2 or a.stride(1) != 1 or w.stride(1) != 1:
        raise _lib.LcvError("gemm_nt: a and w must be 2-D with contiguous rows")
    M, K = a.shape
    N = w.shape[0]
    if w.shape[1] != K:
        raise _lib.LcvError(f"gemm_nt: K mismatch ({K} vs {w.shape[1]})")
    if K % 64:  # rare (toy shapes): zero-pad the contraction dim to the kernel's 64-deep K step
        pad = 64 - K % 64
        a = torch.nn.functional.pad(a, (0, pad))
        w = torch.nn.functional.pad(w, (0, pad))
        K += pad
    K2 = 0
    if a2 is not None:
        _req(a2, BF16, "gemm_nt.a2"); _req(w2, BF16, "gemm_nt.w2")
        K2 = a2.shape[1]
    n_out = N // 2 if epilogue == LCV_EPI_SWIGLU else N
    if out is None:
        out = torch.empty((M, n_out), dtype=F32 if out_f32 else BF16, device=a.device)
    if bias is not None:
        _req(bias, BF16, "gemm_nt.bias")
    if swiglu_aux is not None:
        _req(swiglu_aux, BF16, "gemm_nt.swiglu_aux")
        if epilogue != LCV_EPI_SWIGLU or resid is not None or swiglu_aux.shape != (M, N) or not swiglu_aux.is_contiguous():
            raise _lib.LcvError("gemm_nt: swiglu_aux needs the SwiGLU epilogue and a contiguous [M, N] tensor")
        resid = swiglu_aux          # the C entry point takes it through `resid` (an output under this epilogue)
    C = N
    mod_stride = 0
    if mod is not None:
        _req(mod, F32, "gemm_nt.mod")
        mod_stride = mod.shape[-1]
    call("lcv_gemm_nt", _ptr(a), _ptr(w), _ptr(bias), _ptr(a2), _ptr(w2), _ptr(out), M, N, K, K2,
         a.stride(0), w.stride(0), a2.stride(0) if a2 is not None else 0, w2.stride(0) if w2 is not None else 0,
         out.stride(0), epilogue, 1 if out_f32 else 0, _ptr(resid), _ptr(mod), rows_per_frame, mod_stride,
         gate_idx * C, _stream())
    return out


def linear_f32_smallm(a: torch.Tensor, w: torch.Tensor, bias: Optional[torch.Tensor], act_in: int = 0) -> torch.Tensor:
    """fp32 islands: out[M,N] fp32 = act(a fp32) @ w(bf16)^T + bias."""
    _req(a, F32, "linear_f32_smallm.a"); _req(w, BF16, "linear_f32_smallm.w")
    a = a.contiguous()
    M, K = a.shape
    N = w.shape[0]
    out = torch.empty((M, N), dtype=F32, device=a.device)
    call("lcv_linear_f32_smallm", _ptr(a), _ptr(w.contiguous()), _ptr(bias), _ptr(out), M, N, K, act_in, _stream())
    return out


def lora_down(x: torch.Tensor, A: torch.Tensor, s: float, rpad: int = 64) -> torch.Tensor:
    _req(x, BF16, "lora_down.x"); _req(A, BF16, "lora_down.A")
    M, K = x.shape
    R = A.shape[0]
    h = torch.empty((M, rpad), dtype=BF16, device=x.device)
    call("lcv_lora_down", _ptr(x), _ptr(A.contiguous()), _ptr(h), M, K, R, rpad, x.stride(0), float(s), _stream())
    return h


def swiglu(gate: torch.Tensor, up: torch.Tensor) -> torch.Tensor:
    _req(gate, BF16, "swiglu.gate"); _req(up, BF16, "swiglu.up")
    rows, F = gate.shape
    if gate.stride(0) != up.stride(0) or gate.stride(1) != 1 or up.stride(1) != 1:
        raise _lib.LcvError("swiglu: gate/up must share the row stride")
    out = torch.empty((rows, F), dtype=BF16, device=gate.device)
    call("lcv_swiglu_fwd", _ptr(gate), _ptr(up), _ptr(out), rows, F, gate.stride(0), _stream())
    return out


# ------------------------------------------------------ patch (un)folding ---
def patchify(x: torch.Tensor, kpad: int) -> torch.Tensor:
    _req(x, BF16, "patchify.x")
    B, Cin, T, H, W = x.shape
    x = x.contiguous()
    tok = torch.empty((B, T * (H // 2) * (W // 2), kpad), dtype=BF16, device=x.device)
    call("lcv_patchify", _ptr(x), _ptr(tok), B, Cin, T, H, W, kpad, _stream())
    return tok


def unpatchify(tok: torch.Tensor, Cout: int, T: int, H: int, W: int) -> torch.Tensor:
    B = tok.shape[0]
    tok = tok.contiguous()
    out = torch.empty((B, Cout, T, H, W), dtype=F32, device=tok.device)
    call("lcv_unpatchify", _ptr(tok), _ptr(out), B, Cout, T, H, W, 1 if tok.dtype == F32 else 0, _stream())
    return out


# ---------------------------------------------------------- denoise glue ---
def cfg_euler_step(cond: torch.Tensor, uncond: torch.Tensor, x: torch.Tensor, guidance: float, dt: float,
                   negate: bool = True, zero_star: bool = True) -> None:
    _req(cond, F32, "cfg_euler_step.cond"); _req(uncond, F32, "cfg_euler_step.uncond"); _req(x, F32, "cfg_euler_step.x")
    B = x.shape[0]
    n = x.numel() // B
    ws = torch.empty((B, 256, 2), dtype=F32, device=x.device)      # lcv_hip.h: per-slice partial sums of the zero-star dots
    call("lcv_cfg_euler_step", _ptr(cond.contiguous()), _ptr(uncond.contiguous()), _ptr(x), _ptr(ws), B, n,
         float(guidance), float(dt), 1 if negate else 0, 1 if zero_star else 0, _stream())


def euler_step(v: torch.Tensor, x: torch.Tensor, dt: float, negate: bool = True) -> None:
    _req(v, F32, "euler_step.v"); _req(x, F32, "euler_step.x")
    call("lcv_euler_step", _ptr(v.contiguous()), _ptr(x), x.numel(), float(dt), 1 if negate else 0, _stream())


def fm_noise(x0: torch.Tensor, eps: torch.Tensor, sigma: torch.Tensor) -> torch.Tensor:
    _req(x0, BF16, "fm_noise.x0"); _req(eps, BF16, "fm_noise.eps"); _req(sigma, F32, "fm_noise.sigma")
    B = x0.shape[0]
    out = torch.empty_like(x0)
    call("lcv_fm_noise", _ptr(x0.contiguous()), _ptr(eps.contiguous()), _ptr(sigma), _ptr(out), B,
         x0.numel() // max(B, 1), _stream())
    return out


def fm_mse(pred: torch.Tensor, eps: torch.Tensor, x0: torch.Tensor, Tc: int, need_grad: bool = True):
    _req(pred, F32, "fm_mse.pred"); _req(eps, BF16, "fm_mse.eps"); _req(x0, BF16, "fm_mse.x0")   # the kernel reads raw bf16
    B, C, T, H, W = pred.shape
    if tuple(eps.shape) != (B, C, T - Tc, H, W) or tuple(x0.shape) != tuple(eps.shape):
        raise _lib.LcvError(f"fm_mse: eps {tuple(eps.shape)} / x0 {tuple(x0.shape)} do not match pred {tuple(pred.shape)} "
                            f"with T_cond={Tc}")
    loss = torch.empty((1,), dtype=F32, device=pred.device)
    dpred = torch.empty_like(pred) if need_grad else None
    ws = torch.empty((1024,), dtype=F32, device=pred.device)        # lcv_hip.h: LCV_FM_MSE_BLOCKS per-workgroup partial sums
    call("lcv_fm_mse", _ptr(pred.contiguous()), _ptr(eps.contiguous()), _ptr(x0.contiguous()), _ptr(loss),
         _ptr(dpred), _ptr(ws), B, C, T, Tc, H * W, _stream())
    return loss[0], dpred


def fm_mse_samples(pred: torch.Tensor, eps: torch.Tensor, x0: torch.Tensor, Tc: int) -> torch.Tensor:
    """Per-sample mean((pred[b, :, Tc:] - (eps[b] - x0[b]))^2), fp32 [B], deterministic, no gradient.
    `eps` / `x0` are bf16 [B or 1, C, Tt, H, W]; a leading 1 is shared by every sample without being expanded."""
    _req(pred, F32, "fm_mse_samples.pred"); _req(eps, BF16, "fm_mse_samples.eps"); _req(x0, BF16, "fm_mse_samples.x0")
    B, C, T, H, W = pred.shape
    per = C * (T - Tc) * H * W
    for name, t in (("eps", eps), ("x0", x0)):
        if t.shape[0] not in (1, B) or tuple(t.shape[1:]) != (C, T - Tc, H, W):
            raise _lib.LcvError(f"fm_mse_samples.{name}: shape {tuple(t.shape)} does not match pred {tuple(pred.shape)} with Tc={Tc}")
    eps, x0 = eps.contiguous(), x0.contiguous()
    loss = torch.empty((B,), dtype=F32, device=pred.device)
    ws = torch.empty((B * 256,), dtype=F32, device=pred.device)
    call("lcv_fm_mse_samples", _ptr(pred.contiguous()), _ptr(eps), _ptr(x0), _ptr(loss), _ptr(ws), B, C, T, Tc, H * W,
         per if eps.shape[0] == B and B > 1 else 0, per if x0.shape[0] == B and B > 1 else 0, _stream())
    return loss


# ===================================================================== backward kernels
def _res_grad(dres, like, name):
    """Optional second gradient w.r.t. the norm's input (the residual path of the same block), added inside the kernel."""
    if dres is None:
        return None
    _req(dres, BF16, name)
    if tuple(dres.shape) != tuple(like.shape):
        raise _lib.LcvError(f"{name}: shape {tuple(dres.shape)} does not match x {tuple(like.shape)}")
    return dres.contiguous()


def adaln_modulate_bwd(x, mod, dy, shift_idx, scale_idx, T, eps=1e-6, need_dmod=False, dres=None):
    _req(dy, BF16, "adaln_modulate_bwd.dy")
    B, N, C = x.shape
    dx = torch.empty_like(x)
    dmod = torch.zeros_like(mod) if need_dmod else None
    dres = _res_grad(dres, x, "adaln_modulate_bwd.dres")
    call("lcv_adaln_modulate_bwd", _ptr(x), _ptr(mod), _ptr(dy), _ptr(dx), _ptr(dmod), B, T, N // T, C,
         mod.shape[-1], shift_idx * C, scale_idx * C, eps, _ptr(dres), _stream())
    return dx, dmod


def layernorm_affine_bwd(x, w, dy, eps=1e-6, need_dw=False, dres=None):
    _req(dy, BF16, "layernorm_affine_bwd.dy")
    C = x.shape[-1]
    x = x.contiguous()
    dres = _res_grad(dres, x, "layernorm_affine_bwd.dres")
    wf = w.detach().to(F32).contiguous()
    dx = torch.empty_like(x)
    dw = torch.zeros(C, dtype=F32, device=x.device) if need_dw else None
    db = torch.zeros(C, dtype=F32, device=x.device) if need_dw else None
    call("lcv_layernorm_affine_bwd", _ptr(x), _ptr(wf), _ptr(dy), _ptr(dx), _ptr(dw), _ptr(db), x.numel() // C, C,
         eps, _ptr(dres), _stream())
    return dx, dw, db


def gate_residual_bwd(y, mod, dout, gate_idx, T, need_dmod=False):
    _req(dout, BF16, "gate_residual_bwd.dout")
    B, N, C = y.shape
    dy = torch.empty_like(y)
    dmod = torch.zeros_like(mod) if need_dmod else None
    call("lcv_gate_residual_bwd", _ptr(y), _ptr(mod), _ptr(dout), _ptr(dy), _ptr(dmod), B, T, N // T, C,
         mod.shape[-1], gate_idx * C, _stream())
    return dy, dmod


DW_SLOTS = 256


def qknorm_rope_bwd(q_in, k_in, dq_out, dk_out, dq_in, dk_in, wq, wk, cs, pos_off=0, eps=1e-6, q_scale=1.0,
                    dwq: Optional[torch.Tensor] = None, dwk: Optional[torch.Tensor] = None):
    """dwq / dwk: optional fp32 [128] accumulators (zeroed by the caller) for the norm-weight gradients."""
    ref = q_in if q_in is not None else k_in
    B, N, H, D = ref.shape
    go = dq_out if dq_out is not None else dk_out
    gk = dk_out if dk_out is not None else dq_out
    gi = dq_in if dq_in is not None else dk_in
    for t in (q_in, k_in, dq_out, dk_out, dq_in, dk_in):
        if t is not None and (t.stride(3) != 1 or t.stride(2) != D):
            raise _lib.LcvError("qknorm_rope_bwd: (H, D) must be contiguous")
    # the norm-weight gradients are accumulated into DW_SLOTS rows (token % DW_SLOTS) and added up afterwards: one 128-float
    # target for every token of the call would serialise the kernel on those addresses
    slots = DW_SLOTS if (dwq is not None or dwk is not None) else 1
    sq = torch.zeros((slots, D), dtype=F32, device=ref.device) if dwq is not None else None
    sk = torch.zeros((slots, D), dtype=F32, device=ref.device) if dwk is not None else None
    call("lcv_qknorm_rope_bwd", _ptr(q_in), _ptr(k_in), _ptr(dq_out), _ptr(dk_out), _ptr(dq_in), _ptr(dk_in),
         _ptr(wq), _ptr(wk), _ptr(cs), B, N, H, ref.stride(0), ref.stride(1), go.stride(0), go.stride(1),
         gk.stride(0), gk.stride(1), gi.stride(0), gi.stride(1), pos_off, eps, q_scale, _ptr(sq), _ptr(sk), slots, _stream())
    if dwq is not None:
        dwq.add_(sq.sum(0))
    if dwk is not None:
        dwk.add_(sk.sum(0))


def attention_bwd(q, k, v, o, do, lse, dq, dk, dv, scale, accumulate_kv=False):
    """All [B,N,H,128] views; do must share o's strides; dq/dk/dv written (dk/dv accumulated when asked)."""
    B, Nq, H, D = q.shape
    Nk = k.shape[1]
    if do.stride() != o.stride():
        do = do.contiguous()
        if do.stride() != o.stride():
            raise _lib.LcvError("attention_bwd: dO must share O's strides")
    delta = torch.empty((max(int(_lib.load().lcv_attn_bwd_ws_floats(B, H, Nq, Nk)), 1),), dtype=F32, device=q.device)   # lcv_hip.h: delta_ws
    if PROFILE_BWD is not None:   # bench.py: HIP events on the launch stream around the whole backward of this region
        ev0 = torch.cuda.Event(enable_timing=True); ev1 = torch.cuda.Event(enable_timing=True)
        ev0.record()
    call("lcv_attn_bwd", _ptr(q), _ptr(k), _ptr(v), _ptr(o), _ptr(do), _ptr(lse), _ptr(dq), _ptr(dk), _ptr(dv),
         _ptr(delta), 1 if accumulate_kv else 0, B, H, Nq, Nk,
         q.stride(0), q.stride(1), q.stride(2), k.stride(0), k.stride(1), k.stride(2),
         v.stride(0), v.stride(1), v.stride(2), o.stride(0), o.stride(1), o.stride(2),
         dq.stride(0), dq.stride(1), dq.stride(2), dk.stride(0), dk.stride(1), dk.stride(2),
         dv.stride(0), dv.stride(1), dv.stride(2), float(scale), _stream())
    if PROFILE_BWD is not None:
        ev1.record()
        PROFILE_BWD.append((ev0, ev1, 10.0 * B * H * Nq * Nk * D, Nq, Nk))


def swiglu_bwd(gate, up, dout):
    rows, F = gate.shape
    dg = torch.empty((rows, F), dtype=BF16, device=gate.device)
    du = torch.empty((rows, F), dtype=BF16, device=gate.device)
    call("lcv_swiglu_bwd", _ptr(gate), _ptr(up), _ptr(dout), _ptr(dg), _ptr(du), rows, F, gate.stride(0), _stream())
    return dg, du


def swiglu_bwd_interleaved(gu: torch.Tensor, dout: torch.Tensor) -> torch.Tensor:
    """d[gate | up] (interleaved, [rows, 2F]) from the fused GEMM's saved pre-activations and dh [rows, F]."""
    _req(gu, BF16, "swiglu_bwd_interleaved.gu"); _req(dout, BF16, "swiglu_bwd_interleaved.dout")
    rows, F2 = gu.shape
    if not (gu.is_contiguous() and dout.is_contiguous() and dout.shape == (rows, F2 // 2)):
        raise _lib.LcvError("swiglu_bwd_interleaved: gu [rows, 2F] and dout [rows, F] must be contiguous")
    dgu = torch.empty_like(gu)
    call("lcv_swiglu_bwd_interleaved", _ptr(gu), _ptr(dout), _ptr(dgu), rows, F2 // 2, _stream())
    return dgu


def unpatchify_bwd(dout, Cout, T, H, W):
    _req(dout, F32, "unpatchify_bwd.dout")
    B = dout.shape[0]
    dtok = torch.empty((B, T * (H // 2) * (W // 2), 4 * Cout), dtype=F32, device=dout.device)
    call("lcv_unpatchify_bwd", _ptr(dout), _ptr(dtok), B, Cout, T, H, W, _stream())
    return dtok


def linear_f32_smallm_bwd(dy, w, a, act_in=0):
    _req(dy, F32, "linear_f32_smallm_bwd.dy")
    M, K = a.shape
    N = w.shape[0]
    da = torch.empty((M, K), dtype=F32, device=a.device)
    call("lcv_linear_f32_smallm_bwd", _ptr(dy.contiguous()), _ptr(w.contiguous()), _ptr(a.contiguous()), _ptr(da),
         M, N, K, act_in, _stream())
    return da


def tn_skinny(g, x, R, scale=1.0):
    """out[R, K] fp32 = scale * g[:, :R]^T @ x  (g [M, Rpad] bf16, x [M, K] bf16)."""
    _req(g, BF16, "tn_skinny.g"); _req(x, BF16, "tn_skinny.x")
    M, K = x.shape
    out = torch.empty((R, K), dtype=F32, device=x.device)
    ws_bytes = int(_lib.load().lcv_tn_skinny_ws_bytes(M, K, R))
    ws = torch.empty((max(ws_bytes, 16) // 4,), dtype=F32, device=x.device)     # per-row-group partial sums (caching allocator)
    call("lcv_tn_skinny", _ptr(g), _ptr(x), _ptr(out), M, K, R, g.shape[1], x.stride(0), float(scale), _ptr(ws), ws_bytes,
         _stream())
    return out


# ===================================================================== fused clip + AdamW
class FusedAdamWClip:
    """clip_grad_norm_ + AdamW.step over a fixed parameter list in two launches (norm, update).

    Mirrors `AdamW(lora_params, lr, betas=(0.9, 0.999), weight_decay, eps=1e-8)` +
    `clip_grad_norm_(lora_params, max_norm)` of lora_experiment/scripts/run_lora_tta.py:462-468, 513-514,
    including the bf16 rounding points of the foreach implementation.  `param_groups` is kept so the reference's
    warm-up loop (`for pg in optimizer.param_groups: pg["lr"] = ...`) works unchanged."""
    CHUNK = 2048
    NORM_SLOTS = 64   # partial sums of squares per tensor (csrc/optim.hip)

    def __init__(self, params, lr=2e-4, betas=(0.9, 0.999), weight_decay=0.01, eps=1e-8):
        self.params = [p for p in params]
        if not self.params:
            raise ValueError("optimizer got an empty parameter list")
        dt = self.params[0].dtype
        if dt not in (BF16, F32) or any(p.dtype != dt for p in self.params):
            raise _lib.LcvError("FusedAdamWClip: parameters must be all bf16 or all fp32")
        self.f32 = dt == F32
        self.param_groups = [dict(params=self.params, lr=lr, betas=betas, weight_decay=weight_decay, eps=eps)]
        self.exp_avg = [torch.zeros_like(p) for p in self.params]
        self.exp_avg_sq = [torch.zeros_like(p) for p in self.params]
        self.step_count = 0
        dev = self.params[0].device
        self._ws = torch.zeros(len(self.params) * self.NORM_SLOTS, dtype=F32, device=dev)
        self._norm_coef = torch.zeros(2, dtype=F32, device=dev)
        self._desc = None
        self._desc_key = None
        self._have_coef = False

    def zero_grad(self, set_to_none: bool = True):
        for p in self.params:
            p.grad = None if set_to_none else (p.grad.zero_() if p.grad is not None else None)
        self._have_coef = False

    def _descriptors(self):
        """Device descriptor table over the parameters that received a gradient (torch's AdamW and
        clip_grad_norm_ skip `grad is None` the same way)."""
        sel, grads = [], []
        for i, p in enumerate(self.params):
            if p.grad is None:
                continue
            sel.append(i)
            grads.append(p.grad if p.grad.is_contiguous() else p.grad.contiguous())
        if not sel:
            raise _lib.LcvError("FusedAdamWClip: no parameter has a gradient")
        key = tuple(g.data_ptr() for g in grads) + tuple(sel)
        if self._desc is None or key != self._desc_key:
            rows, chunk = [], 0
            for i, g in zip(sel, grads):
                p, m, v = self.params[i], self.exp_avg[i], self.exp_avg_sq[i]
                rows.append([p.data_ptr(), g.data_ptr(), m.data_ptr(), v.data_ptr(), p.numel(), chunk])
                chunk += (p.numel() + self.CHUNK - 1) // self.CHUNK
            self._desc = torch.tensor(rows, dtype=torch.int64).to(self.params[0].device)
            self._total_chunks = chunk
            self._n_active = len(sel)
            self._desc_key = key
        self._grads = grads  # keep alive until the launch
        return self._desc

    def clip_grad_norm_(self, max_norm: float) -> torch.Tensor:
        d = self._descriptors()
        call("lcv_grad_norm_clip", _ptr(d), self._n_active, self._total_chunks, 1 if self.f32 else 0,
             float(max_norm), _ptr(self._ws), _ptr(self._norm_coef), _stream())
        self._have_coef = True
        return self._norm_coef[0]

    @staticmethod
    def joint_clip_grad_norm_(opts, max_norm: float) -> float:
        """ONE clip over several optimizers' parameters (a bf16 list and an fp32 list under the same
        `clip_grad_norm_(all_params, max_norm)`): per-tensor norms in each tensor's own dtype, the total over their fp32
        stack, coefficient min(max_norm / (total + 1e-6), 1) — torch's rule for mixed dtypes.  Reads the per-tensor
        squares back (one sync per step); used only by the norm + delta tuning option."""
        total_sq = 0.0
        live = [o for o in opts if any(p.grad is not None for p in o.params)]
        for o in live:
            o.clip_grad_norm_(max_norm)                       # fills o._ws with the per-tensor sums of squares
            sq = o._ws[:o._n_active * o.NORM_SLOTS].view(o._n_active, o.NORM_SLOTS).sum(1).sqrt()
            if not o.f32:
                sq = sq.to(BF16).to(F32)                      # a bf16 tensor's norm is a bf16 number
            total_sq += float((sq * sq).sum().item())
        total = total_sq ** 0.5
        coef = min(max_norm / (total + 1e-6), 1.0)
        for o in live:
            o._norm_coef[0] = total
            o._norm_coef[1] = coef
        return total

    def step(self):
        d = self._descriptors()
        g = self.param_groups[0]
        self.step_count += 1
        call("lcv_adamw_step", _ptr(d), self._n_active, self._total_chunks, 1 if self.f32 else 0,
             _ptr(self._norm_coef) if self._have_coef else None, float(g["lr"]), float(g["betas"][0]),
             float(g["betas"][1]), float(g["eps"]), float(g["weight_decay"]), self.step_count, _stream())
        self._have_coef = False
        global PARAM_EPOCH
        PARAM_EPOCH += 1


# ------------------------------------------------------ frame evaluation ---
def gaussian_window11(sigma: float = 1.5):
    """The 11 normalised fp32 taps torchmetrics builds for SSIM (`_gaussian(kernel_size=11, sigma=1.5)`): evaluated
    here in fp32 with the same operation order so the weights are the very same floats."""
    import numpy as np
    dist = np.arange((1 - 11) / 2, (1 + 11) / 2, 1, dtype=np.float32)
    gauss = np.exp(-np.power(dist / np.float32(sigma), 2) / np.float32(2)).astype(np.float32)
    return (gauss / gauss.sum(dtype=np.float32)).astype(np.float32)


def frame_metrics(gen: torch.Tensor, gt: torch.Tensor, data_range: float = 1.0, ssim: Optional[str] = "gaussian11"):
    """gen fp32 [N,H,W,C] in [0,1]; gt fp32 or uint8, same shape.  Returns per-frame (mse fp64 [N], ssim fp64 [N] or None)
    on the host: one launch each, the partial sums added in fp64.  `ssim`: "gaussian11" (torchmetrics defaults),
    "uniform7" (skimage defaults) or None."""
    import ctypes
    import numpy as np
    _req(gen, F32, "frame_metrics.gen")
    if gt.dtype not in (torch.uint8, F32) or not gt.is_cuda:
        raise _lib.LcvError("frame_metrics.gt: fp32 or uint8 GPU tensor expected")
    if gen.shape != gt.shape or gen.dim() != 4:
        raise _lib.LcvError(f"frame_metrics: [N,H,W,C] frames of equal shape expected, got {tuple(gen.shape)} / {tuple(gt.shape)}")
    if ssim not in (None, "gaussian11", "uniform7"):
        raise _lib.LcvError(f"frame_metrics: unknown ssim variant {ssim!r}")
    gen, gt = gen.contiguous(), gt.contiguous()
    N, H, W, C = gen.shape
    win = 7 if ssim == "uniform7" else 11
    if ssim is not None and (H < win or W < win):
        raise _lib.LcvError(f"frame_metrics: a {H}x{W} frame is smaller than the window ({win}x{win})")
    n_sq, n_ss = ctypes.c_int64(0), ctypes.c_int64(0)
    call("lcv_frame_metric_partials", H, W, C, win, ctypes.byref(n_sq), ctypes.byref(n_ss))
    u8 = 1 if gt.dtype == torch.uint8 else 0
    part = torch.empty((N, n_sq.value), dtype=F32, device=gen.device)
    call("lcv_frame_sqerr", _ptr(gen), _ptr(gt), u8, _ptr(part), N, H * W * C, _stream())
    mse = part.double().sum(1).cpu() / float(H * W * C)
    out = None
    if ssim is not None:
        if ssim == "gaussian11":
            taps, cov_norm, clamp = gaussian_window11(), 1.0, 1
        else:
            taps, cov_norm, clamp = np.full(7, 1.0 / 7.0, dtype=np.float32), 49.0 / 48.0, 0
        spart = torch.empty((N, n_ss.value), dtype=F32, device=gen.device)
        call("lcv_frame_ssim", _ptr(gen), _ptr(gt), u8, _ptr(spart), N, H, W, C, taps.ctypes.data, win, cov_norm, clamp,
             (0.01 * data_range) ** 2, (0.03 * data_range) ** 2, _stream())
        out = spart.double().sum(1).cpu() / float((H - win + 1) * (W - win + 1) * C)
    return mse, out


# ------------------------------------------------------- UMT5 text encoder ---
def gather_rows(table: torch.Tensor, ids: torch.Tensor) -> torch.Tensor:
    _req(table, BF16, "gather_rows.table")
    if ids.dtype != torch.int64 or not ids.is_cuda:
        raise _lib.LcvError("gather_rows.ids: int64 GPU tensor expected")
    ids = ids.contiguous().view(-1)
    V, C = table.shape
    out = torch.empty((ids.numel(), C), dtype=BF16, device=table.device)
    call("lcv_gather_rows", _ptr(table.contiguous()), _ptr(ids), _ptr(out), ids.numel(), C, V, _stream())
    return out


def t5_rmsnorm(x: torch.Tensor, w: torch.Tensor, eps: float = 1e-6) -> torch.Tensor:
    _req(x, BF16, "t5_rmsnorm.x"); _req(w, BF16, "t5_rmsnorm.w")
    x = x.contiguous()
    y = torch.empty_like(x)
    C = x.shape[-1]
    call("lcv_t5_rmsnorm", _ptr(x), _ptr(w.contiguous()), _ptr(y), x.numel() // C, C, eps, _stream())
    return y


def geglu_tanh(gate: torch.Tensor, up: torch.Tensor) -> torch.Tensor:
    _req(gate, BF16, "geglu_tanh.gate"); _req(up, BF16, "geglu_tanh.up")
    rows, F = gate.shape
    if gate.stride(0) != up.stride(0) or gate.stride(1) != 1 or up.stride(1) != 1:
        raise _lib.LcvError("geglu_tanh: gate/up must share the row stride")
    out = torch.empty((rows, F), dtype=BF16, device=gate.device)
    call("lcv_geglu_tanh_fwd", _ptr(gate), _ptr(up), _ptr(out), rows, F, gate.stride(0), _stream())
    return out


def t5_attention(qkv: torch.Tensor, H: int, bias_by_dist: torch.Tensor, key_mask: torch.Tensor) -> torch.Tensor:
    """qkv bf16 [B, S, 3*H*64] (q | k | v column blocks); bias_by_dist fp32 [H, 2S-1]; key_mask int32 [B, S].
    Returns [B, S, H*64] bf16."""
    _req(qkv, BF16, "t5_attention.qkv"); _req(bias_by_dist, F32, "t5_attention.bias")
    B, S, W3 = qkv.shape
    inner = H * 64
    if W3 != 3 * inner or qkv.stride(2) != 1:
        raise _lib.LcvError(f"t5_attention: qkv last dim {W3} != 3*{H}*64")
    if tuple(bias_by_dist.shape) != (H, 2 * S - 1) or key_mask.dtype != torch.int32 or tuple(key_mask.shape) != (B, S):
        raise _lib.LcvError("t5_attention: bias_by_dist [H, 2S-1] fp32 and key_mask [B, S] int32 expected")
    out = torch.empty((B, S, inner), dtype=BF16, device=qkv.device)
    base = qkv.data_ptr()
    call("lcv_t5_attention", base, base + inner * 2, base + 2 * inner * 2, _ptr(out), _ptr(bias_by_dist.contiguous()),
         _ptr(key_mask.contiguous()), B, S, H, qkv.stride(1), inner, qkv.stride(0), S * inner, _stream())
    return out


# ------------------------------------------------ full-model TTA (dense backward) ---
def transpose_pad(x: torch.Tensor) -> torch.Tensor:
    """bf16 [M, N] (contiguous rows) -> [N, Mpad] with Mpad = M rounded up to 64, pad columns zero."""
    _req(x, BF16, "transpose_pad.x")
    if x.dim() != 2 or x.stride(1) != 1:
        raise _lib.LcvError("transpose_pad: 2-D tensor with contiguous rows expected")
    M, N = x.shape
    Mpad = (M + 63) // 64 * 64
    out = torch.empty((N, Mpad), dtype=BF16, device=x.device)
    call("lcv_transpose_pad", _ptr(x), _ptr(out), M, N, x.stride(0), Mpad, _stream())
    return out


def rowsum(xT: torch.Tensor, dtype=BF16) -> torch.Tensor:
    _req(xT, BF16, "rowsum.x")
    xT = xT.contiguous()
    out = torch.empty((xT.shape[0],), dtype=dtype, device=xT.device)
    call("lcv_rowsum", _ptr(xT), _ptr(out), xT.shape[0], xT.shape[1], 1 if dtype == F32 else 0, _stream())
    return out


def dense_wgrad(dyT: torch.Tensor, xT: torch.Tensor) -> torch.Tensor:
    """dW [N, K] = dY^T . X from the two transposed, token-padded operands: one NT GEMM over the token axis."""
    return gemm_nt(dyT, xT, None)


def linear_f32_smallm_wgrad(dy: torch.Tensor, a: torch.Tensor, act_in: int, want_db: bool):
    _req(dy, F32, "linear_f32_smallm_wgrad.dy"); _req(a, F32, "linear_f32_smallm_wgrad.a")
    dy, a = dy.contiguous(), a.contiguous()
    M, N = dy.shape
    K = a.shape[1]
    dw = torch.empty((N, K), dtype=BF16, device=dy.device)
    db = torch.empty((N,), dtype=BF16, device=dy.device) if want_db else None
    call("lcv_linear_f32_smallm_wgrad", _ptr(dy), _ptr(a), _ptr(dw), _ptr(db), M, N, K, act_in, _stream())
    return dw, db


def gelu_tanh(x: torch.Tensor) -> torch.Tensor:
    _req(x, BF16, "gelu_tanh.x")
    x = x.contiguous()
    y = torch.empty_like(x)
    call("lcv_gelu_tanh_fwd", _ptr(x), _ptr(y), x.numel(), _stream())
    return y


def gelu_tanh_bwd(x: torch.Tensor, dy: torch.Tensor) -> torch.Tensor:
    _req(x, BF16, "gelu_tanh_bwd.x"); _req(dy, BF16, "gelu_tanh_bwd.dy")
    x, dy = x.contiguous(), dy.contiguous()
    dx = torch.empty_like(x)
    call("lcv_gelu_tanh_bwd", _ptr(x), _ptr(dy), _ptr(dx), x.numel(), _stream())
    return dx


class FusedSGDClip(FusedAdamWClip):
    """clip_grad_norm_ + SGD(momentum=0, weight_decay).step over a parameter list in two launches — the default
    optimizer of full-model TTA (lora_experiment/scripts/run_full_tta.py:138-144, 179-180).  No optimizer state."""

    def __init__(self, params, lr=1e-5, weight_decay=0.01):
        self.params = [p for p in params]
        if not self.params:
            raise ValueError("optimizer got an empty parameter list")
        dt = self.params[0].dtype
        if dt not in (BF16, F32) or any(p.dtype != dt for p in self.params):
            raise _lib.LcvError("FusedSGDClip: parameters must be all bf16 or all fp32")
        self.f32 = dt == F32
        self.param_groups = [dict(params=self.params, lr=lr, weight_decay=weight_decay)]
        self.exp_avg = self.params            # the descriptor table has moment slots; SGD never reads them
        self.exp_avg_sq = self.params
        self.step_count = 0
        dev = self.params[0].device
        self._ws = torch.zeros(len(self.params) * self.NORM_SLOTS, dtype=F32, device=dev)
        self._norm_coef = torch.zeros(2, dtype=F32, device=dev)
        self._desc = None
        self._desc_key = None
        self._have_coef = False

    def step(self):
        d = self._descriptors()
        g = self.param_groups[0]
        self.step_count += 1
        call("lcv_sgd_step", _ptr(d), self._n_active, self._total_chunks, 1 if self.f32 else 0,
             _ptr(self._norm_coef) if self._have_coef else None, float(g["lr"]), float(g["weight_decay"]), _stream())
        self._have_coef = False
        global PARAM_EPOCH
        PARAM_EPOCH += 1
