"""CPU oracle for the text-encoder row (TEST INFRASTRUCTURE — never imported by the product).

The reference encodes every prompt with `transformers.UMT5EncoderModel` (delta_experiment/scripts/common.py:33, 59-64,
228-255: `text_encoder(input_ids, mask).last_hidden_state`, 512 padded tokens).  `transformers` is a third-party
dependency the reference does not pin; the copy installed in this image (5.15.0) IS importable, so this restatement is
pinned directly against it: tests/test_umt5_oracle.py builds random small UMT5 encoders with transformers and requires
`encoder_forward` to reproduce their fp32 output to 1e-5 (and their bf16 output to bf16 noise).

Algorithm (transformers/models/umt5/modeling_umt5.py): token embedding -> N x { RMS norm (fp32 variance, no mean, no
bias) -> self-attention with q.k UNSCALED plus a per-layer learned relative-position bias (32 bidirectional buckets,
max distance 128) plus the additive padding mask -> residual; RMS norm -> gated GELU-tanh MLP (wi_0, wi_1, wo) ->
residual } -> final RMS norm.  `rnd` marks the points where the bf16 model rounds (every linear / elementwise op output).
"""
import math
from typing import Dict

import torch

BF16 = torch.bfloat16


def _id(t):
    return t


def bf16_round(t: torch.Tensor) -> torch.Tensor:
    return t.to(BF16).to(torch.float32)


def relative_position_bucket(relative_position: torch.Tensor, num_buckets: int = 32, max_distance: int = 128) -> torch.Tensor:
    """Bidirectional bucket of (key position - query position); UMT5Attention._relative_position_bucket."""
    nb = num_buckets // 2
    buckets = (relative_position > 0).to(torch.long) * nb
    rp = torch.abs(relative_position)
    max_exact = nb // 2
    is_small = rp < max_exact
    log_ratio = torch.log(rp.float() / max_exact) / math.log(max_distance / max_exact)
    log_ratio = log_ratio * (nb - max_exact)
    large = max_exact + log_ratio.to(torch.long)
    large = torch.min(large, torch.full_like(large, nb - 1))
    return buckets + torch.where(is_small, rp, large)


def rms_norm(x: torch.Tensor, w: torch.Tensor, eps: float, rnd) -> torch.Tensor:
    var = x.pow(2).mean(-1, keepdim=True)
    return rnd(w * rnd(x * torch.rsqrt(var + eps)))


def gelu_new(x: torch.Tensor) -> torch.Tensor:
    return 0.5 * x * (1.0 + torch.tanh(math.sqrt(2.0 / math.pi) * (x + 0.044715 * torch.pow(x, 3.0))))


def encoder_forward(P: Dict[str, torch.Tensor], cfg: Dict, input_ids: torch.Tensor, attention_mask: torch.Tensor,
                    bf16: bool = False) -> torch.Tensor:
    """P: the HF state_dict (fp32 copies); input_ids / attention_mask [B, S].  Returns last_hidden_state [B, S, d_model]."""
    rnd = bf16_round if bf16 else _id
    H, dk, eps = cfg["num_heads"], cfg["d_kv"], cfg.get("layer_norm_epsilon", 1e-6)
    B, S = input_ids.shape
    x = P["shared.weight"][input_ids]                                    # [B, S, C]
    pos = torch.arange(S)
    bucket = relative_position_bucket(pos[None, :] - pos[:, None], cfg.get("relative_attention_num_buckets", 32),
                                      cfg.get("relative_attention_max_distance", 128))          # [S(query), S(key)]
    neg = torch.finfo(torch.bfloat16 if bf16 else torch.float32).min
    mask = (1.0 - attention_mask[:, None, None, :].float()) * neg        # additive, broadcast over heads and queries
    for i in range(cfg["num_layers"]):
        pre = f"encoder.block.{i}.layer."
        h = rms_norm(x, P[pre + "0.layer_norm.weight"], eps, rnd)
        q = rnd(h @ P[pre + "0.SelfAttention.q.weight"].t()).view(B, S, H, dk).transpose(1, 2)
        k = rnd(h @ P[pre + "0.SelfAttention.k.weight"].t()).view(B, S, H, dk).transpose(1, 2)
        v = rnd(h @ P[pre + "0.SelfAttention.v.weight"].t()).view(B, S, H, dk).transpose(1, 2)
        bias = P[pre + "0.SelfAttention.relative_attention_bias.weight"][bucket].permute(2, 0, 1)[None]   # [1, H, S, S]
        s = rnd(q @ k.transpose(2, 3))                                   # no 1/sqrt(d): folded into the trained weights
        s = rnd(rnd(s + bias) + mask)
        p = rnd(torch.softmax(s, dim=-1))
        o = rnd(p @ v).transpose(1, 2).reshape(B, S, H * dk)
        x = rnd(x + rnd(o @ P[pre + "0.SelfAttention.o.weight"].t()))
        h = rms_norm(x, P[pre + "1.layer_norm.weight"], eps, rnd)
        g = rnd(gelu_new(rnd(h @ P[pre + "1.DenseReluDense.wi_0.weight"].t())))
        u = rnd(h @ P[pre + "1.DenseReluDense.wi_1.weight"].t())
        x = rnd(x + rnd(rnd(g * u) @ P[pre + "1.DenseReluDense.wo.weight"].t()))
    return rms_norm(x, P["encoder.final_layer_norm.weight"], eps, rnd)


def make_params(cfg: Dict, seed: int = 0) -> Dict[str, torch.Tensor]:
    """Random bf16-valued weights under the HF state_dict names (synthetic encoders for the GPU parity tests)."""
    g = torch.Generator().manual_seed(seed)
    C, H, dk, F = cfg["d_model"], cfg["num_heads"], cfg["d_kv"], cfg["d_ff"]
    r = lambda *s, std=1.0: (torch.randn(*s, generator=g) * std).to(BF16)
    P = {"shared.weight": r(cfg["vocab_size"], C, std=1.0)}
    for i in range(cfg["num_layers"]):
        pre = f"encoder.block.{i}.layer."
        for n in ("q", "k", "v"):
            P[pre + f"0.SelfAttention.{n}.weight"] = r(H * dk, C, std=C ** -0.5 if n == "v" else (C * dk ** 0.5) ** -0.5 * 2)
        P[pre + "0.SelfAttention.o.weight"] = r(C, H * dk, std=(H * dk) ** -0.5)
        P[pre + "0.SelfAttention.relative_attention_bias.weight"] = r(cfg.get("relative_attention_num_buckets", 32), H, std=1.0)
        P[pre + "0.layer_norm.weight"] = (1.0 + 0.1 * torch.randn(C, generator=g)).to(BF16)
        P[pre + "1.DenseReluDense.wi_0.weight"] = r(F, C, std=C ** -0.5)
        P[pre + "1.DenseReluDense.wi_1.weight"] = r(F, C, std=C ** -0.5)
        P[pre + "1.DenseReluDense.wo.weight"] = r(C, F, std=F ** -0.5)
        P[pre + "1.layer_norm.weight"] = (1.0 + 0.1 * torch.randn(C, generator=g)).to(BF16)
    P["encoder.final_layer_norm.weight"] = (1.0 + 0.1 * torch.randn(C, generator=g)).to(BF16)
    return P
