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


# ------------------------------------------------------- q/k norm + rope ---
def qknorm_rope(q_in: Optional[torch.Tensor], k_in: Optional[torch.Tensor], v_in: Optional[torch.Tensor],
                q_out: Optional[torch.Tensor], k_out: Optional[torch.Tensor], v_out: Optional[torch.Tensor],
                wq: torch.Tensor, wk: torch.Tensor, cs: Optional[torch.Tensor], pos_off: int = 0,
                eps: float = 1e-6) -> None:
    """All tensors are [B, N, H, 128] views with contiguous (H, D); in-place allowed (out is in)."""
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
         ko.stride(0), ko.stride(1), pos_off, eps, _stream())


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
        PROFILE.append((ev0, ev1, 4.0 * B * H * Nq * Nk * D, Nq, Nk))
    return out, lse


# ------------------------------------------------------------------ GEMMs ---
def gemm_nt(a: torch.Tensor, w: torch.Tensor, bias: Optional[torch.Tensor] = None, *,
            a2: Optional[torch.Tensor] = None, w2: Optional[torch.Tensor] = None,
            epilogue: int = LCV_EPI_NONE, out_f32: bool = False, resid: Optional[torch.Tensor] = None,
            mod: Optional[torch.Tensor] = None, gate_idx: int = 0, rows_per_frame: int = 1,
            out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """c[M,N] = a[M,K] @ w[N,K]^T (+ a2 @ w2^T) + bias with a fused epilogue."""
    _req(a, BF16, "gemm_nt.a"); _req(w, BF16, "gemm_nt.w")
    if a.dim() != 2 or w.dim() != 2 or a.stride(1) != 1 or w.stride(1) != 1:
        raise _lib.LcvError("gemm_nt: a and w must be 2-D with contiguous rows")
    M, K = a.shape
    N = w.shape[0]
    if w.shape[1] != K:
        raise _lib.LcvError(f"gemm_nt: K mismatch ({K} vs {w.shape[1]})")
    if K % 64:
        raise _lib.LcvError("gemm_nt: K must be a multiple of 64 (pad on the host)")
    K2 = 0
    if a2 is not None:
        _req(a2, BF16, "gemm_nt.a2"); _req(w2, BF16, "gemm_nt.w2")
        K2 = a2.shape[1]
    n_out = N // 2 if epilogue == LCV_EPI_SWIGLU else N
    if out is None:
        out = torch.empty((M, n_out), dtype=F32 if out_f32 else BF16, device=a.device)
    if bias is not None:
        _req(bias, BF16, "gemm_nt.bias")
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
    ws = torch.empty((B, 2), dtype=F32, device=x.device)
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
    _req(pred, F32, "fm_mse.pred")
    B, C, T, H, W = pred.shape
    loss = torch.empty((1,), dtype=F32, device=pred.device)
    dpred = torch.empty_like(pred) if need_grad else None
    call("lcv_fm_mse", _ptr(pred.contiguous()), _ptr(eps.contiguous()), _ptr(x0.contiguous()), _ptr(loss),
         _ptr(dpred), B, C, T, Tc, H * W, _stream())
    return loss[0], dpred
