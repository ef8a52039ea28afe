"""`LongCatVideoTransformer3DModel` — MI355X-native drop-in for the un-vendored upstream DiT.

Object protocol honoured (every use cited in SURVEY.md §8(b)(i)):
  * `from_pretrained(checkpoint_dir, subfolder="dit", cp_split_hw=[1,1], enable_flashattn2=True,
    torch_dtype=...)`  (delta_experiment/scripts/common.py:71-74)
  * `.config.{patch_size, adaln_tembed_dim, hidden_size, out_channels}`, `.patch_size`, `.x_embedder.proj.weight`,
    `.t_embedder(t_flat, dtype=)`, `.y_embedder`, `.text_tokens_zero_pad`, `.blocks`, `.final_layer(x, t, shape)`,
    `.unpatchify(x, N_t, N_h, N_w)`  (delta_experiment/scripts/run_delta_a.py:146-214)
  * writable `.gradient_checkpointing`, `._gradient_checkpointing_func` (lora_experiment/scripts/run_lora_tta.py:806-811)
  * `__call__(hidden_states=, timestep=, encoder_hidden_states=, encoder_attention_mask=, num_cond_latents=)`
    -> fp32 tensor [B, C_out, T, H, W]  (common.py:476-485)
The outer forward mirrors run_delta_a.py:146-217 step for step.
"""
import json
import os
from types import SimpleNamespace
from typing import Dict, List, Optional

import torch
import torch.nn as nn

from lcv_hip import autograd_ops as A

from .layers import (CaptionEmbedder, FinalLayer_FP32, LongCatSingleStreamBlock, PatchEmbed3D, TimestepEmbedder)

_DEFAULT_CONFIG = dict(
    in_channels=16, out_channels=16, hidden_size=4096, depth=48, num_heads=32, caption_channels=4096,
    mlp_ratio=4, adaln_tembed_dim=512, frequency_embedding_size=256, patch_size=(1, 2, 2),
    text_tokens_zero_pad=False, enable_flashattn2=True, cp_split_hw=(1, 1),
)


class _Config(SimpleNamespace):
    def get(self, k, default=None):
        return getattr(self, k, default)

    def to_dict(self):
        return dict(self.__dict__)


_PACK_PLANS = []   # [(weakref to the mask tensor, its _version, row indices, per-sample lengths)], newest last, at most 8


def _pack_plan(mask: torch.Tensor):
    """(int64 row indices into the flattened [B * L] token rows, [valid tokens per sample]) of a text mask [B, L] (or
    [B, 1, 1, L]).  Cached per mask object + version: entries die with their tensor (a recycled address can never hit)."""
    import weakref
    for ref, ver, idx, lens in _PACK_PLANS:
        if ref() is mask and ver == mask._version:
            return idx, lens
    m2 = mask.reshape(mask.shape[0], -1) != 0
    idx = torch.nonzero(m2.reshape(-1), as_tuple=False).squeeze(1)
    lens = m2.sum(dim=1).tolist()
    _PACK_PLANS[:] = [e for e in _PACK_PLANS if e[0]() is not None][-7:]
    _PACK_PLANS.append((weakref.ref(mask), mask._version, idx, lens))
    return idx, lens


class LongCatVideoTransformer3DModel(nn.Module):
    supports_cond_kv_cache = True      # forward(..., return_kv= / skip_crs_attn= / kv_cache_dict=): the conditioning-frame KV cache

    def __init__(self, device=None, dtype=torch.bfloat16, **cfg):
        super().__init__()
        c = dict(_DEFAULT_CONFIG)
        unknown = set(cfg) - set(c) - {"enable_flashattn3", "enable_xformers", "enable_bsa", "bsa_params", "_class_name",
                                       "_diffusers_version"}
        if unknown:
            raise TypeError(f"unknown DiT config keys: {sorted(unknown)}")
        c.update({k: v for k, v in cfg.items() if k in c})
        c["patch_size"] = tuple(c["patch_size"])
        if c["hidden_size"] // c["num_heads"] != 128:
            raise NotImplementedError("the gfx950 attention kernels are written for head_dim 128")
        if c.get("cp_split_hw") is not None and tuple(c["cp_split_hw"]) != (1, 1):
            raise NotImplementedError("spatial context-parallel split is replaced by frame-axis sequence parallelism")
        self.config = _Config(**c)
        self.patch_size = c["patch_size"]
        self.in_channels, self.out_channels = c["in_channels"], c["out_channels"]
        self.hidden_size, self.num_heads = c["hidden_size"], c["num_heads"]
        self.text_tokens_zero_pad = c["text_tokens_zero_pad"]
        kw = dict(device=device, dtype=dtype)
        self.x_embedder = PatchEmbed3D(self.patch_size, c["in_channels"], c["hidden_size"], **kw)
        self.t_embedder = TimestepEmbedder(c["adaln_tembed_dim"], c["frequency_embedding_size"], **kw)
        self.y_embedder = CaptionEmbedder(c["caption_channels"], c["hidden_size"], **kw)
        self.blocks = nn.ModuleList([
            LongCatSingleStreamBlock(c["hidden_size"], c["num_heads"], c["mlp_ratio"], c["adaln_tembed_dim"],
                                     cp_split_hw=c["cp_split_hw"], **kw)
            for _ in range(c["depth"])
        ])
        num_patch = self.patch_size[0] * self.patch_size[1] * self.patch_size[2]
        self.final_layer = FinalLayer_FP32(c["hidden_size"], num_patch, c["out_channels"], c["adaln_tembed_dim"], **kw)
        self.gradient_checkpointing = False
        self._gradient_checkpointing_func = None
        self._sp_group = None

    # ------------------------------------------------------------------ loading
    @classmethod
    def from_pretrained(cls, checkpoint_dir, subfolder: Optional[str] = None, torch_dtype=torch.bfloat16,
                        cp_split_hw=None, enable_flashattn2=True, device=None, **kwargs):
        path = os.path.join(checkpoint_dir, subfolder) if subfolder else checkpoint_dir
        with open(os.path.join(path, "config.json")) as f:
            cfg = json.load(f)
        cfg = {k: v for k, v in cfg.items() if not k.startswith("_")}
        if cp_split_hw is not None:
            cfg["cp_split_hw"] = tuple(cp_split_hw)
        known = set(_DEFAULT_CONFIG)
        model = cls(device=device or "cpu", dtype=torch_dtype, **{k: v for k, v in cfg.items() if k in known})
        from safetensors.torch import load_file
        # diffusers layout: one `diffusion_pytorch_model.safetensors`, or shards listed by `*.safetensors.index.json`
        index = [f for f in os.listdir(path) if f.endswith(".safetensors.index.json")]
        if index:
            with open(os.path.join(path, index[0])) as f:
                shards = sorted(set(json.load(f)["weight_map"].values()))
            absent = [s for s in shards if not os.path.exists(os.path.join(path, s))]
            if absent:
                raise FileNotFoundError(f"{index[0]} names shards that are not under {path}: {absent}")
        else:
            shards = sorted(f for f in os.listdir(path) if f.endswith(".safetensors"))
        if not shards:
            raise FileNotFoundError(f"no .safetensors weights under {path}")
        state = {}
        for s in shards:
            state.update(load_file(os.path.join(path, s)))
        missing, unexpected = model.load_state_dict(state, strict=False)     # (copy_ casts fp32 shards to torch_dtype)
        if missing:
            raise RuntimeError(f"checkpoint is missing {len(missing)} tensors, e.g. {missing[:5]}")
        return model

    @torch.no_grad()
    def init_synthetic_(self, seed: int = 1234, std: float = 0.02):
        """Random-init weights of the real architecture: N(0, std^2); norm weights 1; LN affine (1, 0)."""
        dev = next(self.parameters()).device
        g = torch.Generator(device=dev).manual_seed(seed)
        for name, p in self.named_parameters():
            if name.endswith("norm.weight") and p.dim() == 1:
                p.fill_(1.0)
            elif name.endswith("pre_crs_attn_norm.bias"):
                p.zero_()
            else:
                p.copy_(torch.randn(p.shape, generator=g, device=dev, dtype=torch.float32).mul_(std))
        return self

    def enable_bsa(self):  # upstream refinement-stage knob: block-sparse attention is out of scope
        raise NotImplementedError("block-sparse attention is not part of the denoise-and-adapt hot path")

    # ------------------------------------------------------------------ pieces
    def unpatchify(self, x, N_t, N_h, N_w):
        pt, ph, pw = self.patch_size
        return A.unpatchify(x, self.out_channels, N_t * pt, N_h * ph, N_w * pw)

    @staticmethod
    def pack_text(encoder_hidden_states, encoder_attention_mask, hidden):
        """Row-major packing of the valid text tokens (run_delta_a.py:180-192: `masked_select` by the mask).  The row indices
        and per-sample lengths of a mask are computed once per mask OBJECT (`_pack_plan`): a 50-step denoise hands the same
        mask to every forward, and `masked_select` + `.tolist()` would be a device -> host sync in each of them (and would
        keep the step out of a hipGraph)."""
        if encoder_attention_mask is not None:
            idx, y_seqlens = _pack_plan(encoder_attention_mask)
            y = encoder_hidden_states.squeeze(1).reshape(-1, hidden).index_select(0, idx).view(1, -1, hidden)
            return y, y_seqlens
        y_seqlens = [encoder_hidden_states.shape[2]] * encoder_hidden_states.shape[0]
        return encoder_hidden_states.squeeze(1).reshape(1, -1, hidden), y_seqlens

    # ------------------------------------------------------------------ sequence parallelism (frame axis)
    def enable_sequence_parallel(self, group=None):
        """Shard the latent frames of every forward over `group` (RCCL over xGMI): per-token work is local, each
        attention layer all-gathers K and V.  `disable_sequence_parallel()` restores the single-GPU path."""
        self._sp_group = (group,)

    def disable_sequence_parallel(self):
        self._sp_group = None
        for b in self.blocks:
            b.attn._sp = None

    def sequence_parallel_sync_grads(self, params) -> None:
        """Under sequence parallelism every rank back-propagates through its own frame shard only, so parameter gradients
        (LoRA adapters, deltas, ...) are partial sums: all-reduce them before the optimizer step."""
        if self._sp_group is None:
            return
        from ..parallel.sequence_parallel import SPContext
        SPContext(1, 1, self._sp_group[0]).all_reduce_grads(params)      # (a context for its collective helpers only)

    def _forward_sp(self, hidden_states, timestep, encoder_hidden_states, encoder_attention_mask, num_cond_latents,
                    kv_cache_dict=None):
        """Token-row-sharded forward (parallel/sequence_parallel.py).  Every rank takes a contiguous run of token rows and runs
        the PLAIN forward on it, presented as a clip of one-row frames: latents [B, C, rows, 2, w], one timestep per row, grid
        (rows, 1, w/2).  Without conditioning frames the whole clip is sharded.  With the conditioning-frame KV cache
        (`kv_cache_dict`: the cond K/V are small and REPLICATED - every rank computed them with the plain path) the NOISE frames
        are sharded and each attention layer attends [cached cond | all-gathered noise].  With the conditioning frames pinned in
        the sequence they are its first rows, possibly split over ranks."""
        from ..parallel.sequence_parallel import SPContext
        B, C, T, H, W = hidden_states.shape
        pt, ph, pw = self.patch_size
        if pt != 1:
            raise NotImplementedError("sequence parallelism assumes temporal patch size 1")
        N_h, N_w = H // ph, W // pw
        sp = SPContext(T, N_h * N_w, self._sp_group[0], rows_per_frame=N_h)
        if len(timestep.shape) == 1:
            timestep = timestep.unsqueeze(1).expand(-1, T)
        rows = hidden_states.reshape(B, C, T * N_h, ph, W)                     # a view: (H) = (N_h, ph), rows of one token each
        ts_rows = timestep.repeat_interleave(N_h, dim=1)                       # a frame's rows share its timestep
        for b in self.blocks:
            b.attn._sp = sp
        try:
            self._sp_group, saved = None, self._sp_group      # the local call below is the plain path on this shard
            if kv_cache_dict is not None:
                ncl_local = num_cond_latents          # cond K/V come from the (replicated) cache; only noise rows are here
            else:                                     # cond frames pinned in the sequence: its first rows, maybe split over ranks
                sp.num_cond_frames = int(num_cond_latents or 0)
                ncl_local = sp.local_units_of_leading_frames(sp.num_cond_frames)
            local = self.forward(rows[:, :, sp.t0:sp.t1].contiguous(), ts_rows[:, sp.t0:sp.t1].contiguous(),
                                 encoder_hidden_states, encoder_attention_mask, ncl_local, kv_cache_dict=kv_cache_dict)
        finally:
            self._sp_group = saved
            for b in self.blocks:
                b.attn._sp = None
        if torch.is_grad_enabled() and local.requires_grad:
            full = sp.gather_frames_autograd(local)                            # [B, C_out, T * N_h, ph, W]
        else:
            full = sp.gather_frames(local)
        return full.reshape(B, full.shape[1], T, H, W)

    # ------------------------------------------------------------------ forward
    def forward(self, hidden_states, timestep, encoder_hidden_states, encoder_attention_mask=None,
                num_cond_latents=0, return_kv=False, kv_cache_dict=None, skip_crs_attn=False, **kwargs):
        if getattr(self, "_sp_group", None) is not None and not return_kv:  # (the cond-frame cache itself is computed replicated)
            return self._forward_sp(hidden_states, timestep, encoder_hidden_states, encoder_attention_mask,
                                    num_cond_latents, kv_cache_dict)
        B, _, T, H, W = hidden_states.shape
        N_t, N_h, N_w = T // self.patch_size[0], H // self.patch_size[1], W // self.patch_size[2]
        if len(timestep.shape) == 1:
            timestep = timestep.unsqueeze(1).expand(-1, N_t)
        dtype = self.x_embedder.proj.weight.dtype
        hidden_states = hidden_states.to(dtype)
        timestep = timestep.to(dtype)
        encoder_hidden_states = encoder_hidden_states.to(dtype)

        x = self.x_embedder(hidden_states)  # [B, N, C]
        t = self.t_embedder(timestep.float().flatten(), dtype=torch.float32).reshape(B, N_t, -1)  # fp32 [B, T, C_t]
        y = self.y_embedder(encoder_hidden_states)
        if self.text_tokens_zero_pad and encoder_attention_mask is not None:
            y = y * encoder_attention_mask[:, None, :, None].to(y.dtype)
            encoder_attention_mask = None        # every (zeroed) token stays in the sequence: the all-ones mask of the reference
        y, y_seqlens = self.pack_text(y, encoder_attention_mask, x.shape[-1])

        kv_out = {} if return_kv else None
        for i, block in enumerate(self.blocks):
            kw = dict(num_cond_latents=num_cond_latents)
            if return_kv:
                x, kv = block(x, y, t, y_seqlens, (N_t, N_h, N_w), return_kv=True, skip_crs_attn=skip_crs_attn, **kw)
                kv_out[i] = kv
            elif kv_cache_dict is not None:
                x = block(x, y, t, y_seqlens, (N_t, N_h, N_w), kv_cache=kv_cache_dict[i], **kw)
            elif torch.is_grad_enabled() and self.gradient_checkpointing and self._gradient_checkpointing_func is not None:
                x = self._gradient_checkpointing_func(block, x, y, t, y_seqlens, (N_t, N_h, N_w), **kw)
            else:
                x = block(x, y, t, y_seqlens, (N_t, N_h, N_w), **kw)
        x = self.final_layer(x, t, (N_t, N_h, N_w))
        x = self.unpatchify(x, N_t, N_h, N_w)
        if return_kv:  # (prediction, per-block K/V): element 0 keeps the output's shape so output hooks (run_delta_c.py:121-133) apply
            return x.to(torch.float32), kv_out
        return x.to(torch.float32)
