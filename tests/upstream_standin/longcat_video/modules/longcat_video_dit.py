"""Stand-in `LongCatVideoTransformer3DModel`: the module tree and call protocol the reference relies on
(SURVEY.md §8(b)(i)), every forward evaluated by `oracle/dit_oracle.py`.  `STANDIN_BREAK=<name>` bends one assumed item so
that the matching guard of tests/first_contact_guards.py can be shown to FAIL."""
import os
import sys
from pathlib import Path
from types import SimpleNamespace

import torch
import torch.nn as nn

sys.path.insert(0, str(Path(__file__).resolve().parents[4]))
from oracle import dit_oracle as O  # noqa: E402

BREAK = os.environ.get("STANDIN_BREAK", "")


def _oracle():
    """The oracle module, bent according to STANDIN_BREAK (a private copy of the few functions involved)."""
    if not BREAK:
        return O
    import types
    M = types.ModuleType("bent_oracle")
    M.__dict__.update(O.__dict__)
    if BREAK == "rope_split":          # t | h | w = 64 | 32 | 32 instead of 44 | 42 | 42
        def rope_angles_3d(grid, head_dim=128, base=10000.0, device=None):
            T, H, W = grid
            dims = (head_dim // 2, head_dim // 4, head_dim // 4)

            def axis(n, dim):
                freqs = 1.0 / (base ** (torch.arange(0, dim, 2)[: dim // 2].float() / dim))
                return torch.outer(torch.arange(n, dtype=torch.float32), freqs).repeat_interleave(2, dim=-1)
            ft, fh, fw = axis(T, dims[0]), axis(H, dims[1]), axis(W, dims[2])
            ang = torch.cat([ft[:, None, None, :].expand(T, H, W, dims[0]), fh[None, :, None, :].expand(T, H, W, dims[1]),
                             fw[None, None, :, :].expand(T, H, W, dims[2])], dim=-1)
            return ang.reshape(T * H * W, head_dim).to(device)
        M.rope_angles_3d = rope_angles_3d
    if BREAK == "sincos":              # timestep features as sin | cos
        def timestep_embedding(t, dim=256, max_period=10000.0):
            e = O.timestep_embedding(t, dim, max_period)
            return torch.cat([e[:, dim // 2:], e[:, :dim // 2]], dim=-1)
        M.timestep_embedding = timestep_embedding
    if BREAK == "rms_eps":
        M.rmsnorm_fp32 = lambda x, w, eps=1e-6, rnd=O._id: O.rmsnorm_fp32(x, w, 1e-2, rnd)
    if BREAK == "gelu_erf":
        def y_embedder(P, y, rnd=O._id):
            h = O.linear(y, P["y_embedder.y_proj.0.weight"], P["y_embedder.y_proj.0.bias"], rnd)
            h = rnd(torch.nn.functional.gelu(h))
            return O.linear(h, P["y_embedder.y_proj.2.weight"], P["y_embedder.y_proj.2.bias"], rnd)
        M.y_embedder = y_embedder
    if BREAK == "cond_sees_all":       # conditioning queries attend every key
        def self_attention(P, pre, x, shape, num_cond_latents, num_heads, rnd=O._id, kv_cache=None, return_kv=False):
            return O.self_attention(P, pre, x, shape, 0, num_heads, rnd, kv_cache, return_kv)
        M.self_attention = self_attention
    if BREAK == "cond_gets_text":      # conditioning tokens receive the text update too
        def cross_attention(P, pre, x, y, y_seqlens, num_cond_latents, shape, num_heads, rnd=O._id):
            return O.cross_attention(P, pre, x, y, y_seqlens, 0, shape, num_heads, rnd)
        M.cross_attention = cross_attention
    if BREAK == "scale_no_plus_one":
        M.modulate_fp32 = lambda x, shift, scale, eps=1e-6, rnd=O._id: rnd(O.layernorm_fp32(x, eps=eps) * scale + shift)
    # functions that call the bent pieces through module globals must be re-bound to the bent module's namespace
    for name in ("self_attention", "cross_attention", "block_forward", "t_embedder", "final_layer", "dit_forward", "ffn",
                 "y_embedder", "modulate_fp32"):
        fn = M.__dict__[name]
        if isinstance(fn, types.FunctionType) and fn.__globals__ is O.__dict__:
            M.__dict__[name] = types.FunctionType(fn.__code__, M.__dict__, fn.__name__, fn.__defaults__, fn.__closure__)
            M.__dict__[name].__kwdefaults__ = fn.__kwdefaults__
    return M


class RMSNorm_FP32(nn.Module):
    def __init__(self, dim, eps=1e-6):
        super().__init__()
        self.eps = eps
        self.weight = nn.Parameter(torch.ones(dim))

    def forward(self, x):
        rnd = O.bf16_round if x.dtype == torch.bfloat16 else O._id
        return _oracle().rmsnorm_fp32(x, self.weight, self.eps, rnd).to(x.dtype)


class _Holder(nn.Module):
    pass


class _TEmbed(nn.Module):
    def __init__(self, dim, freq):
        super().__init__()
        self.frequency_embedding_size = freq
        self.mlp = nn.Sequential(nn.Linear(freq, dim), nn.SiLU(), nn.Linear(dim, dim))

    def forward(self, t, dtype=torch.float32):
        P = {"t_embedder." + k: v for k, v in self.state_dict().items()}
        return _oracle().t_embedder(P, t.float(), self.frequency_embedding_size).to(dtype)


class _YEmbed(nn.Module):
    def __init__(self, cin, c):
        super().__init__()
        act = nn.GELU() if BREAK == "gelu_erf" else nn.GELU(approximate="tanh")
        self.y_proj = nn.Sequential(nn.Linear(cin, c), act, nn.Linear(c, c))

    def forward(self, y):
        P = {"y_embedder." + k: v for k, v in self.state_dict().items()}
        rnd = O.bf16_round if y.dtype == torch.bfloat16 else O._id
        return _oracle().y_embedder(P, y, rnd).to(y.dtype)


class _Block(nn.Module):
    def __init__(self, c, heads, ffn_hidden, ct, idx):
        super().__init__()
        self.idx, self.num_heads = idx, heads
        d = c // heads
        ln_eps = 1e-5 if BREAK == "ln_eps" else 1e-6
        self.adaLN_modulation = nn.Sequential(nn.SiLU(), nn.Linear(ct, 6 * c))
        self.mod_norm_attn = nn.LayerNorm(c, eps=ln_eps, elementwise_affine=False)
        self.mod_norm_ffn = nn.LayerNorm(c, eps=ln_eps, elementwise_affine=False)
        self.pre_crs_attn_norm = nn.LayerNorm(c, eps=ln_eps, elementwise_affine=True)
        self.attn = _Holder()
        self.attn.qkv, self.attn.proj = nn.Linear(c, 3 * c), nn.Linear(c, c)
        self.attn.q_norm, self.attn.k_norm = RMSNorm_FP32(d), RMSNorm_FP32(d)
        self.cross_attn = _Holder()
        self.cross_attn.q_linear, self.cross_attn.kv_linear, self.cross_attn.proj = nn.Linear(c, c), nn.Linear(c, 2 * c), nn.Linear(c, c)
        self.cross_attn.q_norm, self.cross_attn.k_norm = RMSNorm_FP32(d), RMSNorm_FP32(d)
        self.ffn = _Holder()
        self.ffn.w1, self.ffn.w2, self.ffn.w3 = nn.Linear(c, ffn_hidden, bias=False), nn.Linear(ffn_hidden, c, bias=False), nn.Linear(c, ffn_hidden, bias=False)

    def forward(self, x, y, t, y_seqlen, latent_shape, num_cond_latents=None, **kw):
        pre = f"blocks.{self.idx}."
        P = {pre + k: v for k, v in self.state_dict().items()}
        rnd = O.bf16_round if x.dtype == torch.bfloat16 else O._id
        out = _oracle().block_forward(P, pre, x, y, t.float(), y_seqlen, tuple(latent_shape), num_cond_latents, self.num_heads, rnd)
        return out.to(x.dtype)


class _Final(nn.Module):
    def __init__(self, c, n_out, ct):
        super().__init__()
        self.norm_final = nn.LayerNorm(c, eps=1e-6, elementwise_affine=False)
        self.linear = nn.Linear(c, n_out)
        self.adaLN_modulation = nn.Sequential(nn.SiLU(), nn.Linear(ct, 2 * c))


class LongCatVideoTransformer3DModel(nn.Module):
    def __init__(self, in_channels=16, out_channels=16, hidden_size=4096, depth=48, num_heads=32, caption_channels=4096,
                 mlp_ratio=4, adaln_tembed_dim=512, frequency_embedding_size=256, patch_size=(1, 2, 2),
                 text_tokens_zero_pad=False, **unused):
        super().__init__()
        ffn_hidden = O.ffn_hidden_dim(hidden_size, mlp_ratio) + (256 if BREAK == "ffn_width" else 0)
        self.cfg = dict(hidden_size=hidden_size, depth=depth, num_heads=num_heads, in_channels=in_channels, out_channels=out_channels,
                        adaln_tembed_dim=adaln_tembed_dim, caption_channels=caption_channels, patch_size=tuple(patch_size),
                        ffn_hidden=ffn_hidden, frequency_embedding_size=frequency_embedding_size,
                        text_tokens_zero_pad=text_tokens_zero_pad)
        self.config = SimpleNamespace(**self.cfg)
        self.patch_size, self.text_tokens_zero_pad = tuple(patch_size), text_tokens_zero_pad
        self.x_embedder = _Holder()
        self.x_embedder.proj = nn.Conv3d(in_channels, hidden_size, kernel_size=patch_size, stride=patch_size)
        self.t_embedder = _TEmbed(adaln_tembed_dim, frequency_embedding_size)
        self.y_embedder = _YEmbed(caption_channels, hidden_size)
        self.blocks = nn.ModuleList([_Block(hidden_size, num_heads, ffn_hidden, adaln_tembed_dim, i) for i in range(depth)])
        n_patch = patch_size[0] * patch_size[1] * patch_size[2]
        self.final_layer = _Final(hidden_size, n_patch * out_channels, adaln_tembed_dim)
        self.gradient_checkpointing = False

    def forward(self, hidden_states, timestep, encoder_hidden_states, encoder_attention_mask=None, num_cond_latents=0,
                return_kv=False, kv_cache_dict=None, skip_crs_attn=False, **kw):
        P = dict(self.state_dict())
        bf16 = next(self.parameters()).dtype == torch.bfloat16
        return _oracle().dit_forward(P, self.cfg, hidden_states, timestep, encoder_hidden_states, encoder_attention_mask,
                                     num_cond_latents, bf16=bf16, return_kv=return_kv, kv_cache_dict=kv_cache_dict,
                                     skip_crs_attn=skip_crs_attn)
