"""UMT5 text encoder on the HIP kernels.

Drop-in for the object the reference builds with `UMT5EncoderModel.from_pretrained(checkpoint_dir,
subfolder="text_encoder", torch_dtype=bf16)` and calls as `text_encoder(input_ids, mask).last_hidden_state`
(delta_experiment/scripts/common.py:62-64, 250): same constructor entry point, same call, same attribute on the result,
and the module tree uses transformers' parameter names (`shared.weight`, `encoder.block.{i}.layer.0.SelfAttention.{q,k,v,o}`,
`...relative_attention_bias`, `encoder.block.{i}.layer.{0,1}.layer_norm`, `...layer.1.DenseReluDense.{wi_0,wi_1,wo}`,
`encoder.final_layer_norm`) so a Hugging Face checkpoint loads unchanged.

Inference only (the reference never trains it).  Per layer: T5 RMS norm -> ONE fused q|k|v GEMM -> `lcv_t5_attention`
(unscaled q.k + the layer's learned relative-position bias + padding mask, bf16 rounding points of the bf16 model) ->
o GEMM -> residual; RMS norm -> ONE fused wi_0|wi_1 GEMM -> gated GELU-tanh -> wo GEMM -> residual.  The bucket table
(32 bidirectional buckets, max distance 128) is integer work done once per sequence length on the host with the very
operations transformers uses, so the bucket of every (query, key) pair is the same integer.
"""
import json
import math
import os
from types import SimpleNamespace
from typing import Optional

import torch
import torch.nn as nn

from lcv_hip import ops
from .layers import HipLinear

BF16 = torch.bfloat16

_DEFAULT_CONFIG = dict(vocab_size=256384, d_model=4096, d_kv=64, d_ff=10240, num_layers=24, num_heads=64,
                       relative_attention_num_buckets=32, relative_attention_max_distance=128, layer_norm_epsilon=1e-6,
                       feed_forward_proj="gated-gelu")


def relative_position_bucket(relative_position: torch.Tensor, num_buckets: int = 32, max_distance: int = 128) -> torch.Tensor:
    """(key position - query position) -> bucket; bidirectional form of UMT5Attention._relative_position_bucket."""
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


class _T5Norm(nn.Module):
    def __init__(self, dim, eps, device=None, dtype=None):
        super().__init__()
        self.weight = nn.Parameter(torch.ones(dim, device=device, dtype=dtype))
        self.variance_epsilon = eps

    def forward(self, x):
        return ops.t5_rmsnorm(x, self.weight, self.variance_epsilon)


class _SelfAttention(nn.Module):
    def __init__(self, cfg, device=None, dtype=None):
        super().__init__()
        inner = cfg.num_heads * cfg.d_kv
        kw = dict(bias=False, device=device, dtype=dtype)
        self.q, self.k, self.v = HipLinear(cfg.d_model, inner, **kw), HipLinear(cfg.d_model, inner, **kw), HipLinear(cfg.d_model, inner, **kw)
        self.o = HipLinear(inner, cfg.d_model, **kw)
        self.relative_attention_bias = nn.Embedding(cfg.relative_attention_num_buckets, cfg.num_heads, device=device, dtype=dtype)
        self.n_heads = cfg.num_heads
        self._fused = None

    def fused_qkv(self) -> torch.Tensor:
        key = (self.q.weight.data_ptr(), self.k.weight.data_ptr(), self.v.weight.data_ptr(), self.q.weight._version)
        if self._fused is None or self._fused[0] != key:
            self._fused = (key, torch.cat([self.q.weight, self.k.weight, self.v.weight], 0).contiguous())
        return self._fused[1]


class _GatedMLP(nn.Module):
    def __init__(self, cfg, device=None, dtype=None):
        super().__init__()
        kw = dict(bias=False, device=device, dtype=dtype)
        self.wi_0, self.wi_1 = HipLinear(cfg.d_model, cfg.d_ff, **kw), HipLinear(cfg.d_model, cfg.d_ff, **kw)
        self.wo = HipLinear(cfg.d_ff, cfg.d_model, **kw)
        self._fused = None

    def fused_wi(self) -> torch.Tensor:
        key = (self.wi_0.weight.data_ptr(), self.wi_1.weight.data_ptr(), self.wi_0.weight._version)
        if self._fused is None or self._fused[0] != key:
            self._fused = (key, torch.cat([self.wi_0.weight, self.wi_1.weight], 0).contiguous())
        return self._fused[1]


class _Sub(nn.Module):
    """`layer.0` (SelfAttention + layer_norm) or `layer.1` (DenseReluDense + layer_norm), transformers' names."""

    def __init__(self, cfg, kind, device=None, dtype=None):
        super().__init__()
        if kind == 0:
            self.SelfAttention = _SelfAttention(cfg, device, dtype)
        else:
            self.DenseReluDense = _GatedMLP(cfg, device, dtype)
        self.layer_norm = _T5Norm(cfg.d_model, cfg.layer_norm_epsilon, device, dtype)


class _Block(nn.Module):
    def __init__(self, cfg, device=None, dtype=None):
        super().__init__()
        self.layer = nn.ModuleList([_Sub(cfg, 0, device, dtype), _Sub(cfg, 1, device, dtype)])


class _Stack(nn.Module):
    def __init__(self, cfg, device=None, dtype=None):
        super().__init__()
        self.block = nn.ModuleList([_Block(cfg, device, dtype) for _ in range(cfg.num_layers)])
        self.final_layer_norm = _T5Norm(cfg.d_model, cfg.layer_norm_epsilon, device, dtype)


class UMT5EncoderModel(nn.Module):
    def __init__(self, device=None, dtype=BF16, **config):
        super().__init__()
        cfg = dict(_DEFAULT_CONFIG); cfg.update(config)
        self.config = SimpleNamespace(**cfg)
        if self.config.d_kv != 64:
            raise NotImplementedError("lcv_t5_attention is built for d_kv = 64 (UMT5-XXL and every public UMT5 size)")
        if self.config.feed_forward_proj != "gated-gelu":
            raise NotImplementedError("only the gated-GELU feed-forward of UMT5 is built")
        self.shared = nn.Embedding(self.config.vocab_size, self.config.d_model, device=device, dtype=dtype)
        self.encoder = _Stack(self.config, device, dtype)
        self._bucket_cache = {}

    @property
    def dtype(self):
        return self.shared.weight.dtype

    @property
    def device(self):
        return self.shared.weight.device

    def load_state_dict(self, state_dict, strict: bool = True, **kw):
        sd = {k: v for k, v in state_dict.items() if k != "encoder.embed_tokens.weight"}   # tied to shared.weight
        return super().load_state_dict(sd, strict=strict, **kw)

    @classmethod
    def from_pretrained(cls, checkpoint_dir, subfolder: Optional[str] = None, torch_dtype=BF16, device=None, **kwargs):
        path = os.path.join(checkpoint_dir, subfolder) if subfolder else checkpoint_dir
        with open(os.path.join(path, "config.json")) as f:
            raw = json.load(f)
        cfg = {k: raw[k] for k in _DEFAULT_CONFIG if k in raw}
        model = cls(device=device or "cpu", dtype=torch_dtype, **cfg)
        from safetensors.torch import load_file
        shards = sorted(f for f in os.listdir(path) if f.endswith(".safetensors"))
        if not shards:
            raise FileNotFoundError(f"no .safetensors weights under {path}")
        state = {}
        for s in shards:
            state.update(load_file(os.path.join(path, s)))
        missing, unexpected = model.load_state_dict({k: v.to(torch_dtype) for k, v in state.items()}, strict=False)
        if missing:
            raise RuntimeError(f"text-encoder checkpoint is missing {len(missing)} tensors, e.g. {missing[:5]}")
        return model

    @torch.no_grad()
    def init_synthetic_(self, seed: int = 4321):
        """Random weights of the real architecture (no checkpoint offline): N(0, fan_in^-1/2) linears, unit norms."""
        dev = self.device
        g = torch.Generator(device=dev).manual_seed(seed)
        for name, p in self.named_parameters():
            if name.endswith("layer_norm.weight"):
                p.fill_(1.0)
            else:
                std = 1.0 if ("shared" in name or "relative_attention_bias" in name) else p.shape[-1] ** -0.5
                p.copy_(torch.randn(p.shape, generator=g, device=dev, dtype=torch.float32).mul_(std))
        return self

    def _bucket_by_dist(self, S: int) -> torch.Tensor:
        """bucket of (key - query) for every distance -(S-1)..S-1, on the host (integer, exact)."""
        if S not in self._bucket_cache:
            d = torch.arange(-(S - 1), S, dtype=torch.long)
            self._bucket_cache[S] = relative_position_bucket(d, self.config.relative_attention_num_buckets,
                                                             self.config.relative_attention_max_distance).to(self.device)
        return self._bucket_cache[S]

    @torch.no_grad()
    def forward(self, input_ids: torch.Tensor, attention_mask: Optional[torch.Tensor] = None, **kw):
        cfg = self.config
        if input_ids.dim() != 2:
            raise ValueError("input_ids must be [B, S]")
        B, S = input_ids.shape
        ids = input_ids.to(self.device, torch.int64)
        if int(ids.min()) < 0 or int(ids.max()) >= cfg.vocab_size:
            raise IndexError(f"token id outside [0, {cfg.vocab_size})")
        mask = torch.ones((B, S), dtype=torch.int32, device=self.device) if attention_mask is None \
            else attention_mask.to(self.device, torch.int32)
        x = ops.gather_rows(self.shared.weight, ids).view(B * S, cfg.d_model)
        bucket = self._bucket_by_dist(S)
        inner = cfg.num_heads * cfg.d_kv
        for blk in self.encoder.block:
            att, ln0 = blk.layer[0].SelfAttention, blk.layer[0].layer_norm
            h = ln0(x)
            qkv = ops.gemm_nt(h, att.fused_qkv())                                   # [B*S, 3*inner]
            # bias_by_dist[h, d] = relative_attention_bias[bucket(d), h]  (an index lookup: no arithmetic)
            bias = att.relative_attention_bias.weight[bucket].t().float().contiguous()   # [H, 2S-1]
            o = ops.t5_attention(qkv.view(B, S, 3 * inner), cfg.num_heads, bias, mask)
            x = ops.gate_residual(x.view(1, B * S, -1), ops.gemm_nt(o.view(B * S, inner), att.o.weight).view(1, B * S, -1),
                                  None, 0, 1).view(B * S, -1)
            mlp, ln1 = blk.layer[1].DenseReluDense, blk.layer[1].layer_norm
            h = ln1(x)
            gu = ops.gemm_nt(h, mlp.fused_wi())                                      # [B*S, 2*d_ff]: gelu branch | linear branch
            a = ops.geglu_tanh(gu[:, :cfg.d_ff], gu[:, cfg.d_ff:])
            x = ops.gate_residual(x.view(1, B * S, -1), ops.gemm_nt(a, mlp.wo.weight).view(1, B * S, -1),
                                  None, 0, 1).view(B * S, -1)
        x = self.encoder.final_layer_norm(x)
        return SimpleNamespace(last_hidden_state=x.view(B, S, cfg.d_model))
