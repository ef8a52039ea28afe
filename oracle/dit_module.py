"""nn.Module face of the CPU oracle DiT (TEST INFRASTRUCTURE — never imported by the product).

The reference's delta / FiLM / norm-tune / LoRA / full-model code does not call a function, it walks an nn.Module: it reads
`dit.x_embedder.proj.weight.dtype`, calls `dit.t_embedder(t_flat, dtype=)`, `dit.y_embedder`, every `block(x, y, t, y_seqlens,
(N_t, N_h, N_w), num_cond_latents=)`, `dit.final_layer(x, t, grid)`, `dit.unpatchify(x, N_t, N_h, N_w)`, installs forward
(pre-)hooks on `t_embedder`, on blocks, on `block.adaLN_modulation` and on the DiT itself, replaces `attn.qkv` & co. with
`setattr`, and collects norm parameters by attribute path (run_delta_a.py:134-217, run_delta_b.py:175-330, run_delta_c.py:117-165,
run_film_tta.py:146-253, run_norm_tune_tta.py:74-208, run_lora_tta.py:286-382, run_full_tta.py:95-228).

`OracleDiT` exposes exactly that protocol over `oracle/dit_oracle.py`: every parameter is a real nn.Parameter under the
upstream name (so `load_state_dict(make_params(cfg))` works and `state_dict()` keys equal the product's), every linear is a
real nn.Linear invoked through `__call__`, and the arithmetic between them is the oracle's own functions — in fp32 mode
`OracleDiT(cfg, P)(...)` equals `dit_oracle.dit_forward(P, cfg, ..., bf16=False)` exactly (tests/test_delta_golden.py).

Used by tests/golden/make_delta_golden.py to run the REFERENCE's wrappers and loops on CPU and mint fixtures from them.
"""
import types
from typing import Dict

import torch
import torch.nn as nn

from . import dit_oracle as orc


class _Weight(nn.Module):
    """An RMS-norm / LayerNorm parameter holder (the arithmetic lives in dit_oracle.block_forward)."""

    def __init__(self, dim: int, bias: bool = False):
        super().__init__()
        self.weight = nn.Parameter(torch.ones(dim))
        self.bias = nn.Parameter(torch.zeros(dim)) if bias else None


class _SelfAttn(nn.Module):
    def __init__(self, C, D):
        super().__init__()
        self.qkv, self.proj = nn.Linear(C, 3 * C), nn.Linear(C, C)
        self.q_norm, self.k_norm = _Weight(D), _Weight(D)


class _CrossAttn(nn.Module):
    def __init__(self, C, D):
        super().__init__()
        self.q_linear, self.kv_linear, self.proj = nn.Linear(C, C), nn.Linear(C, 2 * C), nn.Linear(C, C)
        self.q_norm, self.k_norm = _Weight(D), _Weight(D)


class _FFN(nn.Module):
    def __init__(self, C, F_):
        super().__init__()
        self.w1, self.w2, self.w3 = nn.Linear(C, F_, bias=False), nn.Linear(F_, C, bias=False), nn.Linear(C, F_, bias=False)


class OracleBlock(nn.Module):
    def __init__(self, cfg: dict, rnd):
        super().__init__()
        C, Ct, H = cfg["hidden_size"], cfg["adaln_tembed_dim"], cfg["num_heads"]
        self.num_heads, self._rnd = H, rnd
        self.adaLN_modulation = nn.Sequential(nn.SiLU(), nn.Linear(Ct, 6 * C))
        self.attn, self.cross_attn = _SelfAttn(C, C // H), _CrossAttn(C, C // H)
        self.pre_crs_attn_norm = _Weight(C, bias=True)
        self.ffn = _FFN(C, cfg["ffn_hidden"])

    def _table(self) -> Dict[str, object]:
        # modules where the reference hooks / replaces them (looked up at call time: `setattr` replacement is seen), parameters
        # for the norms
        a, x = self.attn, self.cross_attn
        return {"adaLN_modulation": self.adaLN_modulation,
                "attn.qkv": a.qkv, "attn.proj": a.proj, "attn.q_norm.weight": a.q_norm.weight, "attn.k_norm.weight": a.k_norm.weight,
                "cross_attn.q_linear": x.q_linear, "cross_attn.kv_linear": x.kv_linear, "cross_attn.proj": x.proj,
                "cross_attn.q_norm.weight": x.q_norm.weight, "cross_attn.k_norm.weight": x.k_norm.weight,
                "pre_crs_attn_norm.weight": self.pre_crs_attn_norm.weight, "pre_crs_attn_norm.bias": self.pre_crs_attn_norm.bias,
                "ffn.w1": self.ffn.w1, "ffn.w2": self.ffn.w2, "ffn.w3": self.ffn.w3}

    def forward(self, x, y, t, y_seqlens, shape, num_cond_latents=None, **kw):
        return orc.block_forward(self._table(), "", x, y, t, y_seqlens, tuple(shape), num_cond_latents, self.num_heads,
                                 self._rnd, kv_cache=kw.get("kv_cache"), return_kv=kw.get("return_kv", False),
                                 skip_crs_attn=kw.get("skip_crs_attn", False))


class _PatchEmbed(nn.Module):
    def __init__(self, cfg, rnd):
        super().__init__()
        self.proj = nn.Conv3d(cfg["in_channels"], cfg["hidden_size"], kernel_size=cfg["patch_size"], stride=cfg["patch_size"])
        self._rnd = rnd

    def forward(self, x):
        return self._rnd(self.proj(x.float()).flatten(2).transpose(1, 2))


class _TimestepEmbedder(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        Ct, Fq = cfg["adaln_tembed_dim"], cfg.get("frequency_embedding_size", 256)
        self.freq_dim = Fq
        self.mlp = nn.Sequential(nn.Linear(Fq, Ct), nn.SiLU(), nn.Linear(Ct, Ct))

    def forward(self, t, dtype=None):
        return self.mlp(orc.timestep_embedding(t, self.freq_dim))


class _CaptionEmbedder(nn.Module):
    def __init__(self, cfg, rnd):
        super().__init__()
        C = cfg["hidden_size"]
        self.y_proj = nn.Sequential(nn.Linear(cfg["caption_channels"], C), nn.GELU(approximate="tanh"), nn.Linear(C, C))
        self._rnd = rnd

    def forward(self, y):
        r = self._rnd
        return r(self.y_proj[2](r(self.y_proj[1](r(self.y_proj[0](y.float()))))))


class _FinalLayer(nn.Module):
    def __init__(self, cfg, rnd):
        super().__init__()
        C, Ct = cfg["hidden_size"], cfg["adaln_tembed_dim"]
        pt, ph, pw = cfg["patch_size"]
        self.adaLN_modulation = nn.Sequential(nn.SiLU(), nn.Linear(Ct, 2 * C))
        self.linear = nn.Linear(C, pt * ph * pw * cfg["out_channels"])
        self._rnd = rnd

    def forward(self, x, t, shape):
        B, N, C = x.shape
        shift, scale = self.adaLN_modulation(t.float()).unsqueeze(2).chunk(2, dim=-1)
        x = orc.modulate_fp32(x.view(B, shape[0], -1, C), shift, scale, rnd=self._rnd).view(B, N, C)
        return self.linear(x.float())


class OracleDiT(nn.Module):
    """`bf16=False` (default): the fp32 ground truth; `bf16=True` rounds at the upstream storage points like
    `dit_forward(bf16=True)` (parameters stay fp32 holders of bf16-representable values)."""

    def __init__(self, cfg: dict, params: Dict[str, torch.Tensor] = None, bf16: bool = False):
        super().__init__()
        self.cfg = dict(cfg)
        rnd = orc.bf16_round if bf16 else orc._id
        self._rnd = rnd
        self.config = types.SimpleNamespace(patch_size=tuple(cfg["patch_size"]), adaln_tembed_dim=cfg["adaln_tembed_dim"],
                                            hidden_size=cfg["hidden_size"], out_channels=cfg["out_channels"],
                                            in_channels=cfg["in_channels"], depth=cfg["depth"], num_heads=cfg["num_heads"])
        self.patch_size = tuple(cfg["patch_size"])
        self.text_tokens_zero_pad = bool(cfg.get("text_tokens_zero_pad", False))
        self.gradient_checkpointing = False
        self._gradient_checkpointing_func = None
        self.x_embedder = _PatchEmbed(cfg, rnd)
        self.t_embedder = _TimestepEmbedder(cfg)
        self.y_embedder = _CaptionEmbedder(cfg, rnd)
        self.blocks = nn.ModuleList([OracleBlock(cfg, rnd) for _ in range(cfg["depth"])])
        self.final_layer = _FinalLayer(cfg, rnd)
        if params is not None:
            missing, unexpected = self.load_state_dict({k: v.float() for k, v in params.items()}, strict=True)
            assert not missing and not unexpected

    def unpatchify(self, x, N_t, N_h, N_w):
        return orc.unpatchify(x, N_t, N_h, N_w, self.patch_size, self.cfg["out_channels"])

    def forward(self, hidden_states, timestep, encoder_hidden_states, encoder_attention_mask=None, num_cond_latents=0, **kw):
        """The outer forward the reference restates at run_delta_a.py:134-217, through the sub-modules' __call__."""
        rnd = self._rnd
        B, _, T, H, W = hidden_states.shape
        pt, ph, pw = self.patch_size
        N_t, N_h, N_w = T // pt, H // ph, W // pw
        if timestep.dim() == 1:
            timestep = timestep.unsqueeze(1).expand(-1, N_t)
        x = self.x_embedder(rnd(hidden_states.float()))
        t = self.t_embedder(rnd(timestep.float()).flatten(), dtype=torch.float32).reshape(B, N_t, -1)
        y = self.y_embedder(rnd(encoder_hidden_states.float()))
        mask = encoder_attention_mask
        if self.text_tokens_zero_pad and mask is not None:
            y = y * mask[:, None, :, None].float()
            mask = (mask * 0 + 1).to(mask.dtype)
        y, y_seqlens = orc.pack_text(y, mask)
        for blk in self.blocks:
            if self.gradient_checkpointing and torch.is_grad_enabled() and self._gradient_checkpointing_func is not None:
                x = self._gradient_checkpointing_func(blk, x, y, t, y_seqlens, (N_t, N_h, N_w), num_cond_latents=num_cond_latents)
            else:
                x = blk(x, y, t, y_seqlens, (N_t, N_h, N_w), num_cond_latents=num_cond_latents)
        x = self.final_layer(x, t, (N_t, N_h, N_w))
        return self.unpatchify(x, N_t, N_h, N_w).float()
