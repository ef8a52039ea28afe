"""Building blocks of the MI355X-native LongCat-Video DiT.

Every module keeps the attribute names the reference reaches for:
linears are real `nn.Linear` subclasses invoked through `__call__`, so the reference's
`setattr` replacement (lora_experiment/scripts/run_lora_tta.py:326-380), `module.forward = hooked`
patching (:137-140) and forward hooks (delta_experiment/scripts/run_film_tta.py:146-163) all take
effect; fused fast paths are used only while the modules they swallow are pristine.
"""
from typing import Optional, Tuple

import torch
import torch.nn as nn

from lcv_hip import autograd_ops as A
from lcv_hip import ops

BF16 = torch.bfloat16


def is_pristine(m: nn.Module, cls) -> bool:
    """True when `m` is exactly `cls`, has no hooks and no patched forward: a fusion may swallow it."""
    return (type(m) is cls and "forward" not in m.__dict__ and not m._forward_hooks
            and not m._forward_pre_hooks and not m._backward_hooks and not getattr(m, "_backward_pre_hooks", {}))


class HipLinear(nn.Linear):
    """nn.Linear whose forward is the gfx950 MFMA GEMM (bf16, fp32 accumulate)."""

    def reset_parameters(self) -> None:  # weights come from a checkpoint or init_synthetic_()
        pass

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        shp = x.shape
        y = A.linear(x.reshape(-1, shp[-1]), self.weight, self.bias)
        return y.view(*shp[:-1], self.out_features)


def _proj_out(proj: "HipLinear", o: torch.Tensor, fuse) -> torch.Tensor:
    """`proj(o)`, or — when the block passes `fuse = (resid, mod, gate_idx, T)` (inference, `proj` pristine) — the
    block's `resid + gate * proj(o)` written by the GEMM's own epilogue: same fp32 arithmetic and rounding points as
    `proj` followed by `gate_residual`, one HBM round trip of the activation less."""
    if fuse is None:
        return proj(o)
    from lcv_hip.lib import LCV_EPI_GATE_RESIDUAL
    resid, mod, gate_idx, T = fuse
    B, N, C = resid.shape
    out = ops.gemm_nt(o.reshape(B * N, -1), proj.weight, proj.bias, epilogue=LCV_EPI_GATE_RESIDUAL,
                      resid=resid.reshape(B * N, C), mod=mod, gate_idx=gate_idx, rows_per_frame=max(N // T, 1))
    return out.view(B, N, C)


class LayerNorm_FP32(nn.LayerNorm):
    """fp32 LayerNorm returning the input dtype (upstream name kept for norm-tuning scripts)."""

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        if self.weight is None:
            raise RuntimeError("the non-affine mod_norm_* layers are fused into adaln_modulate")
        return A.layernorm_affine(x, self.weight, self.bias, self.eps)


class RMSNorm_FP32(nn.Module):
    """Parameter holder for the q/k RMS norms (fused with RoPE into one kernel by the attention modules)."""

    def __init__(self, dim: int, eps: float = 1e-6, device=None, dtype=None):
        super().__init__()
        self.eps = eps
        self.weight = nn.Parameter(torch.ones(dim, device=device, dtype=dtype))


class AdaLNModulation(nn.Sequential):
    """SiLU -> Linear(C_t -> k*C) evaluated in fp32 (the upstream fp32-autocast island).
    Output [B, T, k*C] fp32; forward hooks on this module see and may replace it (FiLM TTA)."""

    def __init__(self, tembed_dim: int, out_dim: int, device=None, dtype=None):
        super().__init__(nn.SiLU(), HipLinear(tembed_dim, out_dim, bias=True, device=device, dtype=dtype))

    def forward(self, t: torch.Tensor) -> torch.Tensor:
        lin = self[1]
        shp = t.shape
        y = A.linear_f32(t.reshape(-1, shp[-1]).float(), lin.weight, lin.bias, act_in=1)
        return y.view(*shp[:-1], lin.out_features)


# ------------------------------------------------------------------ RoPE ---
class RotaryPositionalEmbedding(nn.Module):
    """3-D RoPE tables: head_dim split t | h | w = D-4*(D//6) | 2*(D//6) | 2*(D//6), interleaved pairs,
    base 1e4 [assumed-from-upstream].  Tables are [N, D/2, 2] fp32 (cos, sin), cached per grid."""

    def __init__(self, head_dim: int, cp_split_hw=None):
        super().__init__()
        assert head_dim % 8 == 0
        self.head_dim = head_dim
        self.base = 10000.0
        self.cp_split_hw = cp_split_hw
        self._tables = {}

    def table(self, grid: Tuple[int, int, int], device) -> torch.Tensor:
        key = (tuple(int(g) for g in grid), str(device))
        tab = self._tables.get(key)
        if tab is None:
            T, H, W = key[0]
            D = self.head_dim
            dim_t = D - 4 * (D // 6)
            dim_h = dim_w = 2 * (D // 6)

            def axis(n, dim):
                freqs = 1.0 / (self.base ** (torch.arange(0, dim, 2)[: dim // 2].float() / dim))
                return torch.outer(torch.arange(n, dtype=torch.float32), freqs)  # [n, dim/2]

            ft, fh, fw = axis(T, dim_t), axis(H, dim_h), axis(W, dim_w)
            ang = torch.cat([
                ft[:, None, None, :].expand(T, H, W, dim_t // 2),
                fh[None, :, None, :].expand(T, H, W, dim_h // 2),
                fw[None, None, :, :].expand(T, H, W, dim_w // 2),
            ], dim=-1).reshape(T * H * W, D // 2)
            tab = torch.stack([ang.cos(), ang.sin()], dim=-1).contiguous().to(device)
            self._tables[key] = tab
        return tab


# ------------------------------------------------------------- attention ---
class Attention(nn.Module):
    """3-D spatio-temporal self-attention: qkv -> q/k RMSNorm + RoPE -> flash attention -> proj."""

    def __init__(self, dim: int, num_heads: int, cp_split_hw=None, device=None, dtype=None):
        super().__init__()
        self.dim, self.num_heads = dim, num_heads
        self.head_dim = dim // num_heads
        self.scale = self.head_dim ** -0.5
        self.qkv = HipLinear(dim, dim * 3, bias=True, device=device, dtype=dtype)
        self.q_norm = RMSNorm_FP32(self.head_dim, eps=1e-6, device=device, dtype=dtype)
        self.k_norm = RMSNorm_FP32(self.head_dim, eps=1e-6, device=device, dtype=dtype)
        self.proj = HipLinear(dim, dim, bias=True, device=device, dtype=dtype)
        self.rope_3d = RotaryPositionalEmbedding(self.head_dim, cp_split_hw=cp_split_hw)

    def forward(self, x, shape=None, num_cond_latents=None, return_kv=False, fuse_residual=None):
        B, N, C = x.shape
        H, D = self.num_heads, self.head_dim
        qkv = self.qkv(x).view(B, N, 3, H, D)
        sp = getattr(self, "_sp", None)
        if sp is not None:
            return self._forward_sequence_parallel(qkv, shape, num_cond_latents, sp, fuse_residual)
        cs = self.rope_3d.table(shape, x.device)
        n_cond = 0
        if num_cond_latents is not None and num_cond_latents > 0:
            n_cond = num_cond_latents * (N // shape[0])
        o, kv = A.self_attention(qkv, self.q_norm.weight, self.k_norm.weight, cs, self.scale, n_cond,
                                 self.q_norm.eps, return_kv)
        out = _proj_out(self.proj, o.view(B, N, C), fuse_residual)
        if return_kv:
            return out, kv
        return out

    def _forward_sequence_parallel(self, qkv, shape, num_cond_latents, sp, fuse_residual=None):
        """Token-row-sharded tokens: RoPE at GLOBAL positions (the table of the clip's true grid, addressed by the shard's token
        offset), all-gather of K/V (post-norm, post-RoPE), local-Q x full-KV; `num_cond_latents` counts this rank's LOCAL
        conditioning units (token rows: a prefix of its shard), the global count comes from the SP context.  Differentiable
        (dK / dV are summed over the ranks in the backward)."""
        B, N, _, H, D = qkv.shape
        cs = self.rope_3d.table(sp.grid, qkv.device)
        n_loc = int(num_cond_latents or 0) * sp.S
        n_glob = int(getattr(sp, "num_cond_frames", 0)) * sp.tokens_per_frame
        o = A.sp_self_attention(qkv, self.q_norm.weight, self.k_norm.weight, cs, self.scale, self.q_norm.eps, sp, n_loc, n_glob)
        return _proj_out(self.proj, o.view(B, N, H * D), fuse_residual)

    def forward_with_kv_cache(self, x, shape=None, num_cond_latents=None, kv_cache=None, fuse_residual=None):
        """Denoise step over the noise tokens only; cached (post-norm, post-RoPE) cond K / V lead the keys."""
        B, N, C = x.shape
        H, D = self.num_heads, self.head_dim
        qkv = self.qkv(x).view(B, N, 3, H, D)
        k_c, v_c = kv_cache
        n_c = k_c.shape[1]
        sp = getattr(self, "_sp", None)
        if sp is not None:  # row-sharded noise tokens: global RoPE rows, all-gather of the fresh K/V, cond K/V replicated
            t_c = n_c // sp.tokens_per_frame
            cs = self.rope_3d.table((sp.num_frames + t_c, sp.rows_per_frame, sp.S), x.device)
            q, k, v = qkv[:, :, 0], qkv[:, :, 1], qkv[:, :, 2]
            ops.qknorm_rope(q, k, None, q, k, None, self.q_norm.weight, self.k_norm.weight, cs, n_c + sp.token_offset,
                            self.q_norm.eps, q_scale=ops.log2_qscale(self.scale))
            k_full, v_full = sp.all_gather_kv(k.contiguous(), v.contiguous())
            ex = (lambda t: t if t.shape[0] == B else t.expand(B, -1, -1, -1))
            o, _ = ops.attention(q, torch.cat([ex(k_c), k_full], dim=1), torch.cat([ex(v_c), v_full], dim=1), ops.LN2)
            return _proj_out(self.proj, o.view(B, N, C), fuse_residual)
        T, Hh, Ww = shape
        t_c = n_c // (Hh * Ww)
        cs = self.rope_3d.table((T + t_c, Hh, Ww), x.device)
        o = A.cached_attention(qkv, k_c, v_c, self.q_norm.weight, self.k_norm.weight, cs, self.scale,
                               self.q_norm.eps)
        return _proj_out(self.proj, o.view(B, N, C), fuse_residual)


class MultiHeadCrossAttention(nn.Module):
    """Text cross-attention over the packed valid text tokens (varlen by `kv_seqlen`)."""

    def __init__(self, dim: int, num_heads: int, device=None, dtype=None):
        super().__init__()
        self.dim, self.num_heads = dim, num_heads
        self.head_dim = dim // num_heads
        self.scale = self.head_dim ** -0.5
        self.q_linear = HipLinear(dim, dim, bias=True, device=device, dtype=dtype)
        self.kv_linear = HipLinear(dim, dim * 2, bias=True, device=device, dtype=dtype)
        self.proj = HipLinear(dim, dim, bias=True, device=device, dtype=dtype)
        self.q_norm = RMSNorm_FP32(self.head_dim, eps=1e-6, device=device, dtype=dtype)
        self.k_norm = RMSNorm_FP32(self.head_dim, eps=1e-6, device=device, dtype=dtype)

    def _process_cross_attn(self, x, cond, kv_seqlen, fuse_residual=None):
        B, N, C = x.shape
        H, D = self.num_heads, self.head_dim
        q = self.q_linear(x).view(B, N, H, D)
        kv = self.kv_linear(cond).view(1, -1, 2, H, D)
        o = A.cross_attention(q, kv, self.q_norm.weight, self.k_norm.weight, list(kv_seqlen), self.scale,
                              self.q_norm.eps)
        return _proj_out(self.proj, o.view(B, N, C), fuse_residual)

    def forward(self, x, cond, kv_seqlen, num_cond_latents=None, shape=None, fuse_residual=None):
        if num_cond_latents is None or num_cond_latents == 0:
            return self._process_cross_attn(x, cond, kv_seqlen, fuse_residual)
        if fuse_residual is not None:
            raise ValueError("fuse_residual is only offered without conditioning tokens")
        B, N, C = x.shape
        assert shape is not None, "SHOULD pass in the shape"
        n_cond = num_cond_latents * (N // shape[0])
        out_noise = self._process_cross_attn(x[:, n_cond:], cond, kv_seqlen)
        return A.pad_front_zero(out_noise, n_cond)  # conditioning tokens receive no text update


class FeedForwardSwiGLU(nn.Module):
    def __init__(self, dim: int, hidden_dim: int, multiple_of: int = 256, device=None, dtype=None):
        super().__init__()
        hidden_dim = int(2 * hidden_dim / 3)
        hidden_dim = multiple_of * ((hidden_dim + multiple_of - 1) // multiple_of)
        self.dim, self.hidden_dim = dim, hidden_dim
        self.w1 = HipLinear(dim, hidden_dim, bias=False, device=device, dtype=dtype)
        self.w2 = HipLinear(hidden_dim, dim, bias=False, device=device, dtype=dtype)
        self.w3 = HipLinear(dim, hidden_dim, bias=False, device=device, dtype=dtype)
        self._w13 = None  # [32 gate | 32 up] interleaved copy for the fused SwiGLU epilogue
        self._w13_key = None

    def fused_w13(self) -> Optional[torch.Tensor]:
        """Interleaved copy of (w1, w3) — resident for the job (9 GB of the 288 GB HBM at 48 blocks)."""
        if not (is_pristine(self.w1, HipLinear) and is_pristine(self.w3, HipLinear)):
            return None
        if self.hidden_dim % 32:
            return None
        # Trainable gate / up weights (full-model TTA, or a model nobody froze): the fused optimizers update them through raw
        # pointers, which never bumps `_version`, so the key carries the optimizers' step counter as well - a cached copy is
        # rebuilt after any fused step instead of scoring a model whose FFN is frozen at its pre-TTA values.
        epoch = ops.PARAM_EPOCH if (self.w1.weight.requires_grad or self.w3.weight.requires_grad) else -1
        key = (self.w1.weight.data_ptr(), self.w3.weight.data_ptr(), self.w1.weight._version, self.w3.weight._version, epoch)
        if self._w13 is None or self._w13_key != key:
            F_, K = self.w1.weight.shape
            with torch.no_grad():
                self._w13 = torch.stack([self.w1.weight.view(F_ // 32, 32, K), self.w3.weight.view(F_ // 32, 32, K)],
                                        dim=1).reshape(2 * F_, K).contiguous()
            self._w13_key = key
        return self._w13

    def forward(self, x, fuse_residual=None):
        shp = x.shape
        x2 = x.reshape(-1, shp[-1])
        # under autograd the fused form needs FROZEN gate / up weights (its backward has no dW): LoRA / delta / norm TTA
        pristine = is_pristine(self.w1, HipLinear) and is_pristine(self.w3, HipLinear)   # (a LoRA-wrapped w1 has no .weight)
        trainable = pristine and (self.w1.weight.requires_grad or self.w3.weight.requires_grad)
        w13 = None if (not pristine or (torch.is_grad_enabled() and trainable)) else self.fused_w13()
        if w13 is not None:
            h = A.swiglu_fused(x2, w13)
        else:
            h = A.swiglu(self.w1(x2), self.w3(x2))
        if fuse_residual is not None:
            return _proj_out(self.w2, h.view(*shp[:-1], -1), fuse_residual)
        return self.w2(h).view(*shp[:-1], self.dim)


# -------------------------------------------------------------- embedders ---
class PatchEmbed3D(nn.Module):
    def __init__(self, patch_size, in_chans, embed_dim, device=None, dtype=None):
        super().__init__()
        self.patch_size = tuple(patch_size)
        if self.patch_size != (1, 2, 2):
            raise NotImplementedError("the HIP patchify kernel is written for patch (1, 2, 2)")
        self.proj = nn.Conv3d(in_chans, embed_dim, kernel_size=self.patch_size, stride=self.patch_size,
                              device=device, dtype=dtype)

    def forward(self, x):
        return A.patch_embed(x, self.proj.weight, self.proj.bias)


class TimestepEmbedder(nn.Module):
    """sinusoid(256) -> Linear -> SiLU -> Linear, all fp32; forward hooks on this module see the
    [B*T, C_t] output (delta-A generation hook, run_delta_a.py:117-132)."""

    def __init__(self, t_embed_dim: int, frequency_embedding_size: int = 256, device=None, dtype=None):
        super().__init__()
        self.frequency_embedding_size = frequency_embedding_size
        self.mlp = nn.Sequential(
            HipLinear(frequency_embedding_size, t_embed_dim, bias=True, device=device, dtype=dtype),
            nn.SiLU(),
            HipLinear(t_embed_dim, t_embed_dim, bias=True, device=device, dtype=dtype),
        )

    @staticmethod
    def timestep_embedding(t, dim, max_period=10000):
        return ops.timestep_embedding(t.float(), dim, float(max_period))

    def forward(self, t, dtype=torch.float32):
        t_freq = self.timestep_embedding(t, self.frequency_embedding_size)
        h = A.linear_f32(t_freq, self.mlp[0].weight, self.mlp[0].bias, act_in=0)
        return A.linear_f32(h, self.mlp[2].weight, self.mlp[2].bias, act_in=1)


class CaptionEmbedder(nn.Module):
    def __init__(self, in_channels: int, hidden_size: int, device=None, dtype=None):
        super().__init__()
        self.y_proj = nn.Sequential(
            HipLinear(in_channels, hidden_size, bias=True, device=device, dtype=dtype),
            nn.GELU(approximate="tanh"),
            HipLinear(hidden_size, hidden_size, bias=True, device=device, dtype=dtype),
        )

    def forward(self, caption):
        shp = caption.shape
        x = caption.reshape(-1, shp[-1])
        if torch.is_grad_enabled() and any(p.requires_grad for p in self.y_proj.parameters()):
            # full-model TTA trains the embedder: the activation is its own differentiable op
            h = A.gelu_tanh(A.linear(x, self.y_proj[0].weight, self.y_proj[0].bias))
        else:
            h = A.linear(x, self.y_proj[0].weight, self.y_proj[0].bias, epilogue="gelu_tanh")
        y = A.linear(h, self.y_proj[2].weight, self.y_proj[2].bias)
        return y.view(*shp[:-1], y.shape[-1])


class FinalLayer_FP32(nn.Module):
    def __init__(self, hidden_size, num_patch, out_channels, adaln_tembed_dim, device=None, dtype=None):
        super().__init__()
        self.norm_final = nn.LayerNorm(hidden_size, elementwise_affine=False, eps=1e-6)
        self.linear = HipLinear(hidden_size, num_patch * out_channels, bias=True, device=device, dtype=dtype)
        self.adaLN_modulation = AdaLNModulation(adaln_tembed_dim, 2 * hidden_size, device=device, dtype=dtype)

    def forward(self, x, t, latent_shape):
        T = latent_shape[0]
        mod = self.adaLN_modulation(t)  # [B, T, 2C] fp32
        xm = A.adaln_modulate(x, mod, 0, 1, T, self.norm_final.eps)
        B, N, C = xm.shape
        y = A.linear(xm.view(B * N, C), self.linear.weight, self.linear.bias, out_f32=True)
        return y.view(B, N, -1)


class LongCatSingleStreamBlock(nn.Module):
    def __init__(self, hidden_size, num_heads, mlp_ratio, adaln_tembed_dim, cp_split_hw=None, device=None,
                 dtype=None):
        super().__init__()
        self.hidden_size = hidden_size
        kw = dict(device=device, dtype=dtype)
        self.adaLN_modulation = AdaLNModulation(adaln_tembed_dim, 6 * hidden_size, **kw)
        self.mod_norm_attn = LayerNorm_FP32(hidden_size, eps=1e-6, elementwise_affine=False)
        self.mod_norm_ffn = LayerNorm_FP32(hidden_size, eps=1e-6, elementwise_affine=False)
        self.pre_crs_attn_norm = LayerNorm_FP32(hidden_size, eps=1e-6, elementwise_affine=True, **kw)
        self.attn = Attention(hidden_size, num_heads, cp_split_hw=cp_split_hw, **kw)
        self.cross_attn = MultiHeadCrossAttention(hidden_size, num_heads, **kw)
        self.ffn = FeedForwardSwiGLU(hidden_size, int(hidden_size * mlp_ratio), **kw)

    def forward(self, x, y, t, y_seqlen, latent_shape, num_cond_latents=None, return_kv=False, kv_cache=None,
                skip_crs_attn=False):
        """x [B,N,C]; y [1, sum(y_seqlen), C]; t [B,T,C_t] fp32 (3rd positional arg: run_delta_b.py:188-191)."""
        T = latent_shape[0]
        mod = self.adaLN_modulation(t)  # [B, T, 6C] fp32: shift_msa|scale_msa|gate_msa|shift_mlp|scale_mlp|gate_mlp
        # (x_m, x): under autograd the norm and the residual input leave ONE graph node, whose backward adds the residual
        # path's gradient inside the norm-backward kernel (lcv_hip/autograd_ops.py::_AdaLNForkFn)
        x_m, x = A.adaln_modulate_fork(x, mod, 0, 1, T, self.mod_norm_attn.eps)
        kv = None
        # residual + gate folded into the output projection's GEMM epilogue: inference only, and only while the module
        # whose forward receives the extra argument and the projection it swallows are pristine (hooks / LoRA keep working)
        nograd = not torch.is_grad_enabled()
        fa = (x, mod, 2, T) if (nograd and is_pristine(self.attn, Attention) and is_pristine(self.attn.proj, HipLinear)) else None
        akw = {} if fa is None else {"fuse_residual": fa}
        if kv_cache is not None:
            x_s = self.attn.forward_with_kv_cache(x_m, shape=latent_shape, num_cond_latents=num_cond_latents,
                                                  kv_cache=kv_cache, **akw)
        elif return_kv:
            x_s, kv = self.attn(x_m, shape=latent_shape, num_cond_latents=num_cond_latents, return_kv=True, **akw)
        else:
            x_s = self.attn(x_m, shape=latent_shape, num_cond_latents=num_cond_latents, **akw)
        x = x_s if fa is not None else A.gate_residual(x, x_s, mod, 2, T)
        if not skip_crs_attn:
            ncl = None if kv_cache is not None else num_cond_latents
            fc = (x, None, 0, T) if (nograd and not ncl and is_pristine(self.cross_attn, MultiHeadCrossAttention)
                                     and is_pristine(self.cross_attn.proj, HipLinear)) else None
            ckw = {} if fc is None else {"fuse_residual": fc}
            pn = self.pre_crs_attn_norm
            if torch.is_grad_enabled() and is_pristine(pn, LayerNorm_FP32):
                x_n, x = A.layernorm_affine_fork(x, pn.weight, pn.bias, pn.eps)
            else:
                x_n = pn(x)
            y_s = self.cross_attn(x_n, y, y_seqlen, num_cond_latents=ncl, shape=latent_shape, **ckw)
            x = y_s if fc is not None else A.gate_residual(x, y_s, None, 0, T)
        x_m, x = A.adaln_modulate_fork(x, mod, 3, 4, T, self.mod_norm_ffn.eps)
        ff = (x, mod, 5, T) if (nograd and is_pristine(self.ffn, FeedForwardSwiGLU) and is_pristine(self.ffn.w2, HipLinear)) else None
        x_s = self.ffn(x_m, **({} if ff is None else {"fuse_residual": ff}))
        x = x_s if ff is not None else A.gate_residual(x, x_s, mod, 5, T)
        if return_kv:
            return x, kv
        return x
