"""CPU oracle for the LongCat-Video DiT forward (TEST INFRASTRUCTURE — never imported by the product).

Pure-PyTorch restatement of the un-vendored `meituan-longcat/LongCat-Video` DiT that the
reference drives (unpinned HEAD, `PARTNER_SETUP_GUIDE.md:77-78`).  **Parity unpinned**: the model
source is absent from /root/reference and the reference has no tests for it, so the inner
arithmetic follows (a) what the reference restates itself and (b) the published upstream
algorithm; items marked [assumed-from-upstream] cannot be verified offline.

Followed from the reference (paths relative to /root/reference):
  * outer forward, dtype islands, text packing ........ delta_experiment/scripts/run_delta_a.py:134-217
  * adaLN output layout [shift_msa|scale_msa|gate_msa|shift_mlp|scale_mlp|gate_mlp], SiLU+Linear(512->6C)
    ...................................................... delta_experiment/scripts/run_film_tta.py:5-12,81-82,134-141
  * sub-module names (attn.qkv/proj/q_norm/k_norm, cross_attn.q_linear/kv_linear/proj/q_norm/k_norm,
    pre_crs_attn_norm, ffn.w1/w2/w3) .................... lora_experiment/scripts/run_lora_tta.py:142-168,
                                                           delta_experiment/scripts/run_norm_tune_tta.py:78-96
  * block call signature block(x, y, t, y_seqlens, (N_t,N_h,N_w), num_cond_latents=) ... run_delta_a.py:199-211
  * timestep / num_cond_latents semantics .............. delta_experiment/scripts/common.py:414-489
[assumed-from-upstream]: RoPE axis split (t: D-4*(D//6), h = w = 2*(D//6)), interleaved pairs, base 1e4;
  RMSNorm eps 1e-6 with bf16 `type_as` before the weight multiply; LayerNorm eps 1e-6; t_embedder =
  sinusoid(256, cos|sin) -> Linear -> SiLU -> Linear in fp32; y_embedder Linear -> GELU(tanh) -> Linear;
  cond tokens attend cond tokens only and receive zero cross-attention; SwiGLU w2(silu(w1 x) * w3 x).

`rnd` emulates the bf16 storage points of the upstream modules (`rnd=None` gives the fp32 ground truth).
"""
import math
from typing import Dict, List, Optional, Sequence, Tuple

import torch
import torch.nn.functional as F

BF16 = torch.bfloat16


def _id(t):
    return t


def bf16_round(t: torch.Tensor) -> torch.Tensor:
    return t.to(BF16).to(torch.float32)


def bf16_round_kernel_attention(t: torch.Tensor) -> torch.Tensor:
    """bf16 rounding points as `bf16_round`, and - the attribute below is read by self_attention / cross_attention - the two things
    the HIP attention path does INSIDE an attention that a plain bf16 evaluation does not: q of the self-attention is multiplied by
    head_dim^-0.5 * log2(e) before it is rounded (csrc/elementwise.hip qknorm_rope, `q_scale`), and P is rounded to bf16 against
    the deferred running max (`sdpa_at_kernel_rounding`).  `dit_forward(..., bf16="kernel")` selects it: the checker the GPU
    tests use to hold the whole HIP DiT to the level of bf16 flips instead of the level of two different bf16 evaluations."""
    return t.to(BF16).to(torch.float32)


bf16_round_kernel_attention.kernel_attention = True


# --------------------------------------------------------------------------- RoPE
def rope_angles_3d(grid: Tuple[int, int, int], head_dim: int = 128, base: float = 10000.0, device=None) -> torch.Tensor:
    """[T*H*W, head_dim] angles, each frequency repeated for its (2i, 2i+1) pair; axis order t | h | w.
    Built on the CPU (so the table is the same whichever device evaluates the oracle), then moved."""
    T, H, W = grid
    dim_t = head_dim - 4 * (head_dim // 6)
    dim_h = 2 * (head_dim // 6)
    dim_w = 2 * (head_dim // 6)

    def axis(n, dim):
        freqs = 1.0 / (base ** (torch.arange(0, dim, 2)[: dim // 2].float() / dim))
        pos = torch.arange(n, dtype=torch.float32)
        return torch.outer(pos, freqs).repeat_interleave(2, dim=-1)  # [n, dim]

    ft, fh, fw = axis(T, dim_t), axis(H, dim_h), axis(W, dim_w)
    ang = torch.cat([
        ft[:, None, None, :].expand(T, H, W, dim_t),
        fh[None, :, None, :].expand(T, H, W, dim_h),
        fw[None, None, :, :].expand(T, H, W, dim_w),
    ], dim=-1)
    return ang.reshape(T * H * W, head_dim).to(device)


def rope_cos_sin_table(grid: Tuple[int, int, int], head_dim: int = 128) -> torch.Tensor:
    """[N, head_dim/2, 2] fp32 (cos, sin) per pair — the table layout the HIP kernel consumes."""
    ang = rope_angles_3d(grid, head_dim)[:, 0::2]
    return torch.stack([ang.cos(), ang.sin()], dim=-1).contiguous()


def rotate_half(x: torch.Tensor) -> torch.Tensor:
    x = x.reshape(*x.shape[:-1], -1, 2)
    x1, x2 = x.unbind(dim=-1)
    return torch.stack((-x2, x1), dim=-1).flatten(-2)


def apply_rope(x: torch.Tensor, ang: torch.Tensor, rnd=_id) -> torch.Tensor:
    """x [B, H, N, D] (any float dtype); ang [N, D]."""
    xf = x.float()
    cos, sin = ang.cos()[None, None], ang.sin()[None, None]
    return rnd(xf * cos + rotate_half(xf) * sin)


# --------------------------------------------------------------------------- norms
def rmsnorm_fp32(x: torch.Tensor, w: torch.Tensor, eps: float = 1e-6, rnd=_id) -> torch.Tensor:
    xf = x.float()
    n = rnd(xf * torch.rsqrt(xf.pow(2).mean(-1, keepdim=True) + eps))  # .type_as(x)
    return rnd(n * w.float())


def layernorm_fp32(x, w=None, b=None, eps: float = 1e-6):
    return F.layer_norm(x.float(), (x.shape[-1],), None if w is None else w.float(),
                        None if b is None else b.float(), eps)


def modulate_fp32(x, shift, scale, eps: float = 1e-6, rnd=_id):
    """x [B, T, S, C]; shift/scale [B, T, 1, C] fp32."""
    return rnd(layernorm_fp32(x, eps=eps) * (scale + 1) + shift)


def linear(x, w, b=None, rnd=_id):
    y = x.float() @ w.float().t()
    if b is not None:
        y = y + b.float()
    return rnd(y)


def _lin(P, name, x, rnd=_id):
    """The linear `name` of the parameter table P.  P[name] may be a callable — an nn.Module standing in for the
    (weight, bias) pair (oracle/dit_module.py) — so that forward hooks and `setattr` replacement act on the oracle exactly
    as they do on the upstream model (run_lora_tta.py:333, run_film_tta.py:146-163)."""
    f = P.get(name)
    if callable(f):
        return rnd(f(x))
    return linear(x, P[name + ".weight"], P.get(name + ".bias"), rnd)


def sdpa(q, k, v, scale, rnd=_id):
    """q [B,H,Nq,D], k/v [B,H,Nk,D] -> [B,H,Nq,D]; softmax in fp32.  Heads are walked in groups when the score
    matrix would not fit (full-size clips evaluated with this file on the GPU box's card: tests/test_gpu_denoise_parity.py)."""
    B, H, Nq, _ = q.shape
    per_head = B * Nq * k.shape[2] * 4
    step = max(1, min(H, int((6 << 30) // max(per_head, 1))))
    outs = []
    for h0 in range(0, H, step):
        s = (q[:, h0:h0 + step].float() @ k[:, h0:h0 + step].float().transpose(-1, -2)) * scale
        p = torch.softmax(s, dim=-1)
        del s
        outs.append(rnd(p @ v[:, h0:h0 + step].float()))
        del p
    return outs[0] if len(outs) == 1 else torch.cat(outs, dim=1)


def sdpa_at_kernel_rounding(q, k, v, scale, tile=64, group=32, thr=6.0, row0=0):
    """The forward attention kernels' arithmetic restated tile by tile (checker for tests/test_gpu_kernels.py and
    test_gpu_fullsize.py, not a second definition of the op; csrc/attn_fwd.hip:284-345, attn_fwd_pipe.hip, attn_fwd_w64.hip:233-256):
    keys in tiles of 64; scores in log2 units; the running max of a query row is raised only when SOME row of its 32-row group sees a
    tile maximum more than 2^6 above its own running max (and on the first tile), then O and l are rescaled; P = exp2(s - m_run) is
    rounded to bf16 as the MFMA operand of P V while the row sum takes the unrounded fp32 P; O / l is stored in bf16.
    `row0`: index of q's first row inside the launch (groups are aligned to the launch's row 0), for checks on a slice of rows.
    Against this the kernels differ only by fp32 summation order and the bf16 flips that follow from it."""
    c = scale * 1.4426950408889634
    B, H, Nq, D = q.shape
    Nk = k.shape[2]
    qf, kf, vf = q.float(), k.float(), v.float()
    dev = q.device
    gid = (torch.arange(Nq, device=dev) + row0) // group
    gid = gid - gid.min()
    ng = int(gid.max()) + 1
    m = torch.zeros(B, H, Nq, device=dev)
    l = torch.zeros(B, H, Nq, device=dev)
    o = torch.zeros(B, H, Nq, D, device=dev)
    for t, k0 in enumerate(range(0, Nk, tile)):
        s = (qf @ kf[:, :, k0:k0 + tile].transpose(-1, -2)) * c
        mx = s.amax(dim=-1)
        if t == 0:
            m_new = mx
        else:
            grew = ((mx - m) > thr).float()
            over = torch.zeros(B, H, ng, device=dev).scatter_reduce_(2, gid.expand(B, H, Nq), grew, "amax", include_self=True) > 0
            m_new = torch.where(over[:, :, gid], torch.maximum(m, mx), m)
        alpha = torch.exp2(m - m_new) if t else torch.zeros_like(m)
        m = m_new
        pt = torch.exp2(s - m.unsqueeze(-1))
        l = l * alpha + pt.sum(dim=-1)
        o = o * alpha.unsqueeze(-1) + bf16_round(pt) @ vf[:, :, k0:k0 + tile]
    return bf16_round(o / l.unsqueeze(-1))


def sdpa_backward_at_kernel_rounding(q, k, v, d_o, scale, scale_inside=True, o=None, lse=None):
    """dq, dk, dv of softmax(q k^T scale) v at the backward kernels' rounding points: P and dS are bf16 MFMA operands, results
    stored in bf16.  `o` [B,H,Nq,D] and `lse` [B,H,Nq] (natural log): the forward's outputs, which the kernels take as INPUTS
    (delta = rowsum(dO * O), P = exp(s - lse)); without them the fp32 softmax and its bf16-rounded output stand in.
    `scale_inside`: the general-scale forms round dS = P (dP - delta) * scale (csrc/attn_bwd.hip:214-223, 451-454); the unit-scale
    forms the DiT's self-attention takes (q already in log2 units, scale = ln 2) round dS' = P (dP - delta) and multiply dQ / dK by
    the scale afterwards (attn_bwd_dq2.hip, attn_bwd_dkv2.hip) - with a scale that is not a power of two those are different
    roundings."""
    qf, kf, vf, gf = q.float(), k.float(), v.float(), d_o.float()
    s = (qf @ kf.transpose(-1, -2)) * scale
    p = torch.softmax(s, dim=-1) if lse is None else torch.exp(s - lse.float().unsqueeze(-1))
    of = bf16_round(p @ vf) if o is None else o.float()
    delta = (gf * of).sum(dim=-1, keepdim=True)
    dv = bf16_round(p).transpose(-1, -2) @ gf
    dp = gf @ vf.transpose(-1, -2)
    if scale_inside:
        ds = bf16_round(p * (dp - delta) * scale)
        dq, dk = ds @ kf, ds.transpose(-1, -2) @ qf
    else:
        ds = bf16_round(p * (dp - delta))
        dq, dk = (ds @ kf) * scale, (ds.transpose(-1, -2) @ qf) * scale
    return bf16_round(dq), bf16_round(dk), bf16_round(dv)


# --------------------------------------------------------------------------- block pieces
def _attend(q, k, v, scale, rnd):
    if getattr(rnd, "kernel_attention", False):
        return sdpa_at_kernel_rounding(q, k, v, scale)
    return sdpa(q, k, v, scale, rnd)


def _rope_q(q, ang, scale, rnd):
    """RoPE on q.  Kernel mode: the product folds head_dim^-0.5 * log2(e) (as an fp32 constant) into q BEFORE the bf16 rounding and
    then attends with scale ln 2; returns (q, scale to attend with)."""
    if getattr(rnd, "kernel_attention", False):
        c = float(torch.tensor(scale * 1.4426950408889634, dtype=torch.float32))
        return rnd(apply_rope(q, ang, _id) * c), math.log(2.0)
    return apply_rope(q, ang, rnd), scale


def self_attention(P: Dict[str, torch.Tensor], pre: str, x, shape, num_cond_latents, num_heads, rnd=_id,
                   kv_cache=None, return_kv=False):
    B, N, C = x.shape
    D = C // num_heads
    qkv = _lin(P, pre + "qkv", x, rnd)
    qkv = qkv.view(B, N, 3, num_heads, D).permute(2, 0, 3, 1, 4)  # [3,B,H,N,D]
    q, k, v = qkv.unbind(0)
    q = rmsnorm_fp32(q, P[pre + "q_norm.weight"], rnd=rnd)
    k = rmsnorm_fp32(k, P[pre + "k_norm.weight"], rnd=rnd)
    scale = D ** -0.5
    if kv_cache is not None:
        # KV-cached denoise step: cached cond K (pre-RoPE) / V in front, positions continue after them
        k_c, v_c = kv_cache
        if k_c.shape[0] != B:  # one cached conditioning clip shared by the CFG pair
            k_c, v_c = k_c.expand(B, -1, -1, -1), v_c.expand(B, -1, -1, -1)
        T, Hh, Ww = shape
        n_c = k_c.shape[2]
        t_c = n_c // (Hh * Ww)
        ang = rope_angles_3d((T + t_c, Hh, Ww), D, device=x.device)
        k_full = torch.cat([k_c.float(), k.float()], dim=2)
        v_full = torch.cat([v_c.float(), v.float()], dim=2)
        k_full = apply_rope(k_full, ang, rnd)
        q, sc = _rope_q(q, ang[n_c:], scale, rnd)
        o = _attend(q, k_full, v_full, sc, rnd)
    else:
        kv = (k.clone(), v.clone()) if return_kv else None
        ang = rope_angles_3d(tuple(shape), D, device=x.device)
        (q, sc), k = _rope_q(q, ang, scale, rnd), apply_rope(k, ang, rnd)
        if num_cond_latents is not None and num_cond_latents > 0:
            nc = num_cond_latents * (N // shape[0])
            o_c = _attend(q[:, :, :nc], k[:, :, :nc], v[:, :, :nc], sc, rnd)
            o_n = _attend(q[:, :, nc:], k, v, sc, rnd)    # its own launch: row groups count from the first noisy row
            o = torch.cat([o_c, o_n], dim=2)
        else:
            o = _attend(q, k, v, sc, rnd)
    o = o.transpose(1, 2).reshape(B, N, C)
    out = _lin(P, pre + "proj", o, rnd)
    if return_kv and kv_cache is None:
        return out, kv
    return out


def cross_attention(P, pre, x, y, y_seqlens: Sequence[int], num_cond_latents, shape, num_heads, rnd=_id):
    """x [B,N,C]; y [1, sum(y_seqlens), C] packed valid text tokens."""
    B, N, C = x.shape
    D = C // num_heads
    nc = 0
    if num_cond_latents is not None and num_cond_latents > 0:
        nc = num_cond_latents * (N // shape[0])
    xn = x[:, nc:]
    q = _lin(P, pre + "q_linear", xn, rnd).view(B, N - nc, num_heads, D)
    kv = _lin(P, pre + "kv_linear", y, rnd).view(1, -1, 2, num_heads, D)
    k, v = kv.unbind(2)
    q = rmsnorm_fp32(q, P[pre + "q_norm.weight"], rnd=rnd)
    k = rmsnorm_fp32(k, P[pre + "k_norm.weight"], rnd=rnd)
    outs = []
    off = 0
    for b in range(B):
        L = int(y_seqlens[b])
        kb, vb = k[0, off:off + L], v[0, off:off + L]
        off += L
        o = _attend(q[b].transpose(0, 1)[None], kb.transpose(0, 1)[None], vb.transpose(0, 1)[None], D ** -0.5, rnd)
        outs.append(o[0].transpose(0, 1).reshape(N - nc, C))
    o = torch.stack(outs, 0)
    o = _lin(P, pre + "proj", o, rnd)
    if nc > 0:
        o = torch.cat([torch.zeros(B, nc, C, dtype=o.dtype, device=o.device), o], dim=1)
    return o


def ffn(P, pre, x, rnd=_id):
    g = _lin(P, pre + "w1", x, rnd)
    u = _lin(P, pre + "w3", x, rnd)
    h = rnd(rnd(F.silu(g)) * u)
    return _lin(P, pre + "w2", h, rnd)


def adaln_table(P, pre, t):
    """fp32 island: SiLU -> Linear(C_t -> k*C) on t [B, T, C_t] fp32.  P[pre without the dot] may be the module itself
    (oracle/dit_module.py), so FiLM's forward hooks on `adaLN_modulation` see its output (run_film_tta.py:146-163)."""
    f = P.get(pre[:-1])
    if callable(f):
        return f(t.float())
    return F.silu(t.float()) @ P[pre + "1.weight"].float().t() + P[pre + "1.bias"].float()


def block_forward(P, pre, x, y, t, y_seqlens, shape, num_cond_latents, num_heads, rnd=_id, kv_cache=None,
                  return_kv=False, skip_crs_attn=False, mod_add=None):
    B, N, C = x.shape
    T = shape[0]
    mod = adaln_table(P, pre + "adaLN_modulation.", t)
    if mod_add is not None:  # FiLM: a [6C] correction added to the adaLN output (run_film_tta.py:148-151)
        mod = mod + mod_add.view(1, 1, -1).to(mod.dtype)
    mod = mod.unsqueeze(2)  # [B,T,1,6C]
    sh_msa, sc_msa, g_msa, sh_mlp, sc_mlp, g_mlp = mod.chunk(6, dim=-1)
    x_m = modulate_fp32(x.view(B, T, -1, C), sh_msa, sc_msa, rnd=rnd).view(B, N, C)
    res = self_attention(P, pre + "attn.", x_m, shape, num_cond_latents, num_heads, rnd, kv_cache, return_kv)
    kv = None
    if return_kv and kv_cache is None:
        x_s, kv = res
    else:
        x_s = res
    x = rnd(x.float() + (g_msa * x_s.float().view(B, T, -1, C)).view(B, N, C))
    if not skip_crs_attn:
        ncl = None if kv_cache is not None else num_cond_latents
        xn = rnd(layernorm_fp32(x, P[pre + "pre_crs_attn_norm.weight"], P[pre + "pre_crs_attn_norm.bias"]))
        x = rnd(x.float() + cross_attention(P, pre + "cross_attn.", xn, y, y_seqlens, ncl, shape, num_heads, rnd).float())
    x_m = modulate_fp32(x.view(B, T, -1, C), sh_mlp, sc_mlp, rnd=rnd).view(B, N, C)
    x_s = ffn(P, pre + "ffn.", x_m, rnd)
    x = rnd(x.float() + (g_mlp * x_s.float().view(B, T, -1, C)).view(B, N, C))
    if return_kv and kv_cache is None:
        return x, kv
    return x


# --------------------------------------------------------------------------- embedders / head
def timestep_embedding(t: torch.Tensor, dim: int = 256, max_period: float = 10000.0) -> torch.Tensor:
    half = dim // 2
    freqs = torch.exp(-math.log(max_period) * torch.arange(half, dtype=torch.float32) / half).to(t.device)
    args = t[:, None].float() * freqs[None]
    return torch.cat([torch.cos(args), torch.sin(args)], dim=-1)


def t_embedder(P, t_flat: torch.Tensor, freq_dim: int = 256) -> torch.Tensor:
    e = timestep_embedding(t_flat, freq_dim)
    h = e @ P["t_embedder.mlp.0.weight"].float().t() + P["t_embedder.mlp.0.bias"].float()
    return F.silu(h) @ P["t_embedder.mlp.2.weight"].float().t() + P["t_embedder.mlp.2.bias"].float()


def y_embedder(P, y, rnd=_id):
    h = linear(y, P["y_embedder.y_proj.0.weight"], P["y_embedder.y_proj.0.bias"], rnd)
    h = rnd(F.gelu(h, approximate="tanh"))
    return linear(h, P["y_embedder.y_proj.2.weight"], P["y_embedder.y_proj.2.bias"], rnd)


def x_embedder(P, x, patch, rnd=_id):
    w = P["x_embedder.proj.weight"].float()
    y = F.conv3d(x.float(), w, P["x_embedder.proj.bias"].float(), stride=patch)
    return rnd(y.flatten(2).transpose(1, 2))


def final_layer(P, x, t, shape, rnd=_id):
    B, N, C = x.shape
    T = shape[0]
    mod = adaln_table(P, "final_layer.adaLN_modulation.", t).unsqueeze(2)
    shift, scale = mod.chunk(2, dim=-1)
    x = modulate_fp32(x.view(B, T, -1, C), shift, scale, rnd=rnd).view(B, N, C)
    return x.float() @ P["final_layer.linear.weight"].float().t() + P["final_layer.linear.bias"].float()


def unpatchify(x, N_t, N_h, N_w, patch, c_out):
    B = x.shape[0]
    pt, ph, pw = patch
    x = x.view(B, N_t, N_h, N_w, pt, ph, pw, c_out)
    x = x.permute(0, 7, 1, 4, 2, 5, 3, 6)
    return x.reshape(B, c_out, N_t * pt, N_h * ph, N_w * pw)


def pack_text(y_emb: torch.Tensor, mask: Optional[torch.Tensor]):
    """[B,1,L,C] + [B,L] -> ([1, sum, C], seqlens) by row-major masked_select (run_delta_a.py:180-192)."""
    B, _, L, C = y_emb.shape
    if mask is None:
        return y_emb.squeeze(1).reshape(1, -1, C), [L] * B
    m = mask.reshape(B, L)
    y = y_emb.squeeze(1).masked_select(m.unsqueeze(-1) != 0).view(1, -1, C)
    return y, m.sum(dim=1).tolist()


def dit_forward(P: Dict[str, torch.Tensor], cfg: dict, hidden_states, timestep, encoder_hidden_states,
                encoder_attention_mask=None, num_cond_latents=0, bf16: bool = True, t_delta=None,
                return_kv: bool = False, kv_cache_dict=None, skip_crs_attn: bool = False, adapters: Optional[dict] = None):
    """Full forward following run_delta_a.py:134-217.  cfg: depth, num_heads, patch_size, out_channels,
    text_tokens_zero_pad.

    `return_kv` / `skip_crs_attn` / `kv_cache_dict` are the three switches the pipeline's conditioning-frame KV
    cache uses [assumed-from-upstream]: the clean conditioning latents pass once at t = 0 without text
    cross-attention and every block keeps its (K pre-RoPE, V); a denoise step then runs over the noise tokens
    only, with the cached cond K/V in front of the keys and RoPE positions continuing after them.

    `adapters` — where the reference's delta / FiLM wrappers touch this forward (each entry optional; per-block lists hold a
    tensor or None per block):
      "block_t"      [depth] x [C_t]  added to the `t` a block receives ............ run_delta_b.py:294-298 (timestep target)
      "block_hidden" [depth] x [C]    added to the hidden stream after a block ..... run_delta_b.py:311-318 (hidden target)
      "final_hidden" [C]              added in front of the final layer ............ run_delta_b.py:321-324 (training forward only)
      "film"         [depth] x [6C]   added to the block's adaLN output ............ run_film_tta.py:146-151
      "out_delta"    [C_out]          added to the prediction ...................... run_delta_c.py:159-163"""
    rnd = bf16_round_kernel_attention if bf16 == "kernel" else (bf16_round if bf16 else _id)
    B, _, T, H, W = hidden_states.shape
    pt, ph, pw = cfg["patch_size"]
    N_t, N_h, N_w = T // pt, H // ph, W // pw
    if timestep.dim() == 1:
        timestep = timestep.unsqueeze(1).expand(-1, N_t)
    hs = rnd(hidden_states.float())
    ts = rnd(timestep.float())  # timestep.to(dtype): the bf16 round trip of sigma*1000
    y = rnd(encoder_hidden_states.float())
    x = x_embedder(P, hs, (pt, ph, pw), rnd)
    t = t_embedder(P, ts.flatten()).reshape(B, N_t, -1)
    if t_delta is not None:  # delta-A: run_delta_a.py:168
        t = t + t_delta.unsqueeze(0).unsqueeze(0)
    y = y_embedder(P, y, rnd)
    mask = encoder_attention_mask
    if cfg.get("text_tokens_zero_pad", False) and mask is not None:
        y = y * mask[:, None, :, None].float()
        mask = (mask * 0 + 1).to(mask.dtype)
    y, y_seqlens = pack_text(y, mask)
    kv_out = {} if return_kv else None
    ad = adapters or {}

    def _at(key, i):
        lst = ad.get(key)
        return None if lst is None else lst[i]

    for i in range(cfg["depth"]):
        kvc = None if kv_cache_dict is None else kv_cache_dict[i]
        bt = _at("block_t", i)
        t_i = t if bt is None else t + bt.float().view(1, 1, -1)
        r = block_forward(P, f"blocks.{i}.", x, y, t_i, y_seqlens, (N_t, N_h, N_w), num_cond_latents,
                          cfg["num_heads"], rnd, kv_cache=kvc, return_kv=return_kv, skip_crs_attn=skip_crs_attn,
                          mod_add=_at("film", i))
        if return_kv:
            x, kv_out[i] = r
        else:
            x = r
        bh = _at("block_hidden", i)
        if bh is not None:  # `hidden_states + delta.to(hidden_states.dtype)`: one rounding of delta, one of the sum
            x = rnd(x + rnd(bh.float()).view(1, 1, -1))
    if ad.get("final_hidden") is not None:
        x = rnd(x + rnd(ad["final_hidden"].float()).view(1, 1, -1))
    x = final_layer(P, x, t, (N_t, N_h, N_w), rnd)
    out = unpatchify(x, N_t, N_h, N_w, (pt, ph, pw), cfg["out_channels"]).float()
    if ad.get("out_delta") is not None:
        out = out + ad["out_delta"].float().view(1, -1, 1, 1, 1)
    if return_kv:
        return out, kv_out
    return out


# --------------------------------------------------------------------------- synthetic weights
def make_params(cfg: dict, seed: int = 1234, std: float = 0.02) -> Dict[str, torch.Tensor]:
    """Random-init weights N(0, std^2) in bf16 (norm weights 1, LN affine (1,0)); names follow the
    drop-in module's state_dict.  SURVEY 8(d) synthetic-input recipe."""
    g = torch.Generator().manual_seed(seed)
    C, Ct, Cin, Cout = cfg["hidden_size"], cfg["adaln_tembed_dim"], cfg["in_channels"], cfg["out_channels"]
    Cy, Fh, D = cfg["caption_channels"], cfg["ffn_hidden"], cfg["hidden_size"] // cfg["num_heads"]
    pt, ph, pw = cfg["patch_size"]
    P = {}

    def w(name, *shape):
        P[name] = (torch.randn(*shape, generator=g) * std).to(BF16)

    w("x_embedder.proj.weight", C, Cin, pt, ph, pw); w("x_embedder.proj.bias", C)
    w("t_embedder.mlp.0.weight", Ct, cfg.get("frequency_embedding_size", 256)); w("t_embedder.mlp.0.bias", Ct)
    w("t_embedder.mlp.2.weight", Ct, Ct); w("t_embedder.mlp.2.bias", Ct)
    w("y_embedder.y_proj.0.weight", C, Cy); w("y_embedder.y_proj.0.bias", C)
    w("y_embedder.y_proj.2.weight", C, C); w("y_embedder.y_proj.2.bias", C)
    for i in range(cfg["depth"]):
        p = f"blocks.{i}."
        w(p + "adaLN_modulation.1.weight", 6 * C, Ct); w(p + "adaLN_modulation.1.bias", 6 * C)
        w(p + "attn.qkv.weight", 3 * C, C); w(p + "attn.qkv.bias", 3 * C)
        w(p + "attn.proj.weight", C, C); w(p + "attn.proj.bias", C)
        P[p + "attn.q_norm.weight"] = torch.ones(D, dtype=BF16)
        P[p + "attn.k_norm.weight"] = torch.ones(D, dtype=BF16)
        w(p + "cross_attn.q_linear.weight", C, C); w(p + "cross_attn.q_linear.bias", C)
        w(p + "cross_attn.kv_linear.weight", 2 * C, C); w(p + "cross_attn.kv_linear.bias", 2 * C)
        w(p + "cross_attn.proj.weight", C, C); w(p + "cross_attn.proj.bias", C)
        P[p + "cross_attn.q_norm.weight"] = torch.ones(D, dtype=BF16)
        P[p + "cross_attn.k_norm.weight"] = torch.ones(D, dtype=BF16)
        P[p + "pre_crs_attn_norm.weight"] = torch.ones(C, dtype=BF16)
        P[p + "pre_crs_attn_norm.bias"] = torch.zeros(C, dtype=BF16)
        w(p + "ffn.w1.weight", Fh, C); w(p + "ffn.w2.weight", C, Fh); w(p + "ffn.w3.weight", Fh, C)
    w("final_layer.adaLN_modulation.1.weight", 2 * C, Ct); w("final_layer.adaLN_modulation.1.bias", 2 * C)
    w("final_layer.linear.weight", pt * ph * pw * Cout, C); w("final_layer.linear.bias", pt * ph * pw * Cout)
    return P


def ffn_hidden_dim(hidden_size: int, mlp_ratio: float = 4.0, multiple_of: int = 256) -> int:
    h = int(2 * int(hidden_size * mlp_ratio) / 3)
    return multiple_of * ((h + multiple_of - 1) // multiple_of)


def small_config(hidden_size=256, depth=2, num_heads=2, caption_channels=64) -> dict:
    return dict(hidden_size=hidden_size, depth=depth, num_heads=num_heads, in_channels=16, out_channels=16,
                adaln_tembed_dim=64, caption_channels=caption_channels, patch_size=(1, 2, 2),
                ffn_hidden=ffn_hidden_dim(hidden_size), frequency_embedding_size=256, text_tokens_zero_pad=False)
