"""delta-TTA ("AdaSteer") wrappers on the MI355X DiT: a learned vector added to the timestep embedding (A: one global
delta; B: one per block group, optionally on the hidden stream, partial dim, block subset) or to the output (C).

Class / method / argument names follow delta_experiment/scripts/run_delta_a.py:88-305, run_delta_b.py:99-421 and
run_delta_c.py:82-246; every wrapper and loop here is checked on the GPU against fixtures minted from those classes
themselves (tests/golden/make_delta_golden.py, tests/test_gpu_delta_golden.py).  Where the reference re-implements the DiT's
outer forward for training and installs hooks for generation, this build uses the same hook points for both (the reference's
own Series-25 check shows the two are equivalent) — with the one difference the reference has between them kept: delta-B's
`delta_final` acts in the training forward only (run_delta_b.py:321-324 vs :175-212).  The DiT calls `t_embedder`, each block and itself through `__call__`, gradient checkpointing stays on the
DiT (`dit.gradient_checkpointing`), and the gradient reaches delta through the fp32 adaLN island
(`lcv_linear_f32_smallm_bwd`) and the modulation-table gradients of `lcv_adaln_modulate_bwd` / `lcv_gate_residual_bwd`.
The optimizer is the fused clip + AdamW in its fp32 form (delta lives in fp32: run_delta_a.py:104).
"""
import math
from typing import Dict, List, Optional

import torch
import torch.nn as nn
import torch.nn.functional as F

from lcv_hip.ops import FusedAdamWClip

from .early_stopping import AnchoredEarlyStopper
from .inner_loop import _OneVideo, _fm_loss, run_adaptation
from .lora import _parse_target_blocks


class _HookedWrapper(nn.Module):
    def __init__(self, dit: nn.Module):
        super().__init__()
        self.dit = dit
        for p in self.dit.parameters():
            p.requires_grad = False
        self._hooks: list = []

    @property
    def config(self):
        return self.dit.config

    def remove_from_dit(self):
        for h in self._hooks:
            h.remove()
        self._hooks = []

    # the early stopper may score its anchors through the conditioning-frame KV cache (tta/early_stopping.py::_AnchorSet): the
    # hooks below act on the cache pass and on the cached pass exactly as they do on the pinned sequence, which is also how the
    # continuation runs them (`apply_to_dit()` around `pipe.generate_vc`, whose first call is the cache pass)
    supports_cond_kv_cache = True
    _CACHE_KW = ("return_kv", "skip_crs_attn", "kv_cache_dict")

    def forward(self, hidden_states, timestep, encoder_hidden_states, encoder_attention_mask=None,
                num_cond_latents=0, **kwargs):
        cache_kw = {k: kwargs[k] for k in self._CACHE_KW if k in kwargs}
        self.apply_to_dit()
        try:
            return self.dit(hidden_states=hidden_states, timestep=timestep, encoder_hidden_states=encoder_hidden_states,
                            encoder_attention_mask=encoder_attention_mask, num_cond_latents=num_cond_latents, **cache_kw)
        finally:
            self.remove_from_dit()


class DeltaAWrapper(_HookedWrapper):
    """One delta in R^{C_t} added to the timestep embedding (t_embedder output [B*T, C_t])."""

    def __init__(self, dit: nn.Module, adaln_tembed_dim: int = 512):
        super().__init__(dit)
        self.delta = nn.Parameter(torch.zeros(adaln_tembed_dim))

    def apply_to_dit(self):
        self.remove_from_dit()
        delta = self.delta
        self._hooks.append(self.dit.t_embedder.register_forward_hook(
            lambda _m, _i, out: out + delta.unsqueeze(0).to(out.dtype)))


class DeltaBWrapper(_HookedWrapper):
    """Per-group deltas: group(i) = min(i // ceil(L/G), G-1); partial-dim deltas are zero-padded to the full width."""

    def __init__(self, dit: nn.Module, num_groups: int = 4, adaln_tembed_dim: int = 512, hidden_size: int = 4096,
                 delta_target: str = "timestep", delta_dim: Optional[int] = None, target_blocks: str = "all"):
        super().__init__(dit)
        if delta_target not in ("timestep", "hidden"):
            raise ValueError(f"Unknown delta_target: {delta_target}")
        self.num_groups = num_groups
        self.num_blocks = len(dit.blocks)
        self.delta_target = delta_target
        self.target_block_indices = _parse_target_blocks(target_blocks, self.num_blocks)
        full_dim = adaln_tembed_dim if delta_target == "timestep" else hidden_size
        self._full_dim = full_dim
        self._partial_dim = delta_dim if delta_dim is not None else full_dim
        self.deltas = nn.ParameterList([nn.Parameter(torch.zeros(self._partial_dim)) for _ in range(num_groups)])
        # the final-layer delta of the "hidden" target is sized by `delta_dim` itself (run_delta_b.py:149): without --delta-dim
        # the reference's constructor raises TypeError from torch.zeros(None), and so does this one
        self.delta_final = nn.Parameter(torch.zeros(delta_dim)) if delta_target == "hidden" else None
        per = math.ceil(self.num_blocks / num_groups)
        self.block_to_group = [min(i // per, num_groups - 1) for i in range(self.num_blocks)]

    def _pad_delta(self, dv: torch.Tensor) -> torch.Tensor:
        if dv.shape[0] >= self._full_dim:
            return dv
        return F.pad(dv, (0, self._full_dim - dv.shape[0]))

    def apply_to_dit(self):
        """The GENERATION hooks (run_delta_b.py:175-212): per-block deltas only.  The reference trains `delta_final` in its own
        forward (:321-324) but never applies it while the video is generated; `forward()` below adds it, this method does not
        (tests/golden/delta_wrappers.pt: `pred_gen` != `pred_train` for the hidden target)."""
        self._install(training=False)

    def forward(self, hidden_states, timestep, encoder_hidden_states, encoder_attention_mask=None,
                num_cond_latents=0, **kwargs):
        cache_kw = {k: kwargs[k] for k in self._CACHE_KW if k in kwargs}
        self._install(training=True)
        try:
            return self.dit(hidden_states=hidden_states, timestep=timestep, encoder_hidden_states=encoder_hidden_states,
                            encoder_attention_mask=encoder_attention_mask, num_cond_latents=num_cond_latents, **cache_kw)
        finally:
            self.remove_from_dit()

    def _install(self, training: bool):
        self.remove_from_dit()
        pad = self._pad_delta
        for i, block in enumerate(self.dit.blocks):
            if self.target_block_indices is not None and i not in self.target_block_indices:
                continue
            dv = self.deltas[self.block_to_group[i]]
            if self.delta_target == "timestep":
                def pre(_m, args, dv=dv):
                    args = list(args)
                    args[2] = args[2] + pad(dv).unsqueeze(0).unsqueeze(0).to(args[2].dtype)  # `t` is args[2]
                    return tuple(args)
                self._hooks.append(block.register_forward_pre_hook(pre))
            else:
                def post(_m, _a, out, dv=dv):
                    e = pad(dv).unsqueeze(0).unsqueeze(0)
                    if isinstance(out, tuple):
                        return (out[0] + e.to(out[0].dtype),) + out[1:]
                    return out + e.to(out.dtype)
                self._hooks.append(block.register_forward_hook(post))
        if training and self.delta_target == "hidden" and self.delta_final is not None:
            df = self.delta_final

            def final_pre(_m, args, df=df):
                args = list(args)
                args[0] = args[0] + pad(df).unsqueeze(0).unsqueeze(0).to(args[0].dtype)
                return tuple(args)
            self._hooks.append(self.dit.final_layer.register_forward_pre_hook(final_pre))


class DeltaCWrapper(_HookedWrapper):
    """Output bias: pred + delta_out.view(1, C_out, 1, 1, 1); no gradient flows into the DiT."""

    def __init__(self, dit: nn.Module, mode: str = "per_channel", out_channels: int = 16):   # run_delta_c.py:91-96
        super().__init__(dit)
        self.mode = mode
        if mode != "per_channel":
            raise ValueError(f"Unknown mode: {mode}. Use 'per_channel'.")
        self.delta_out = nn.Parameter(torch.zeros(out_channels))

    def apply_to_dit(self):
        self.remove_from_dit()
        d = self.delta_out

        def _hook(_m, _i, out):  # the KV-cache pass returns (prediction, kv): run_delta_c.py:125-131
            if isinstance(out, tuple):
                return (out[0] + d.view(1, -1, 1, 1, 1).to(out[0].dtype),) + out[1:]
            return out + d.view(1, -1, 1, 1, 1).to(out.dtype)

        self._hooks.append(self.dit.register_forward_hook(_hook))

    def forward(self, hidden_states, timestep, encoder_hidden_states, encoder_attention_mask=None,
                num_cond_latents=0, **kwargs):
        cache_kw = {k: kwargs[k] for k in self._CACHE_KW if k in kwargs}
        with torch.no_grad():
            pred = self.dit(hidden_states=hidden_states, timestep=timestep, encoder_hidden_states=encoder_hidden_states,
                            encoder_attention_mask=encoder_attention_mask, num_cond_latents=num_cond_latents, **cache_kw)
        if isinstance(pred, tuple):          # the cache pass: (prediction, per-block K / V)
            return (pred[0] + self.delta_out.view(1, -1, 1, 1, 1).to(pred[0].dtype),) + pred[1:]
        return pred + self.delta_out.view(1, -1, 1, 1, 1).to(pred.dtype)


class FiLMAdapterWrapper(_HookedWrapper):
    """Per-block-group additive corrections to the adaLN output [shift_msa | scale_msa | gate_msa | shift_mlp | scale_mlp |
    gate_mlp] (delta_experiment/scripts/run_film_tta.py:78-176): forward hooks on every block's `adaLN_modulation`; the
    gradient reaches the corrections through the modulation-table gradients of the adaLN / gate-residual kernels."""

    def __init__(self, dit: nn.Module, num_groups: int = 4, hidden_size: int = 4096, film_mode: str = "full"):
        super().__init__(dit)
        dims = {"full": 6, "shift_scale": 4, "scale_only": 2}
        if film_mode not in dims:
            raise ValueError(f"Unknown film_mode: {film_mode}")
        self.num_groups, self.num_blocks = num_groups, len(dit.blocks)
        self.hidden_size, self.film_mode = hidden_size, film_mode
        self.correction_dim = dims[film_mode] * hidden_size
        self.corrections = nn.ParameterList([nn.Parameter(torch.zeros(self.correction_dim)) for _ in range(num_groups)])

    def _get_group_idx(self, block_idx: int) -> int:
        return block_idx * self.num_groups // self.num_blocks

    def _expand_correction(self, corr: torch.Tensor) -> torch.Tensor:
        C = self.hidden_size
        if self.film_mode == "full":
            return corr
        z = torch.zeros(C, device=corr.device, dtype=corr.dtype)
        if self.film_mode == "scale_only":   # [scale_msa, scale_mlp]
            return torch.cat([z, corr[:C], z, z, corr[C:], z])
        return torch.cat([corr[:C], corr[C:2 * C], z, corr[2 * C:3 * C], corr[3 * C:], z])  # shift_scale

    def apply_to_dit(self):
        self.remove_from_dit()
        for i, blk in enumerate(self.dit.blocks):
            corr = self.corrections[self._get_group_idx(i)]
            self._hooks.append(blk.adaLN_modulation.register_forward_hook(
                lambda _m, _i, out, corr=corr: out + self._expand_correction(corr).view(1, 1, -1).to(out.dtype)))

    def reset_corrections(self):
        for c in self.corrections:
            c.data.zero_()


def _optimize(wrapper: nn.Module, params: List[nn.Parameter], per_param_clip: bool, cond_latents, train_latents,
              prompt_embeds, prompt_mask, num_steps, lr, device, dtype, early_stopper, train_latents_variants):
    """AdamW(0.9, 0.999, wd 0.01, eps 1e-15), clip at 1.0, no warm-up (run_delta_a.py:224-305 and its siblings) on the
    shared engine.  Clipping comes in the three shapes the reference scripts use: per parameter (delta-B,
    run_delta_b.py:386-388), one global norm, or — bf16 norm weights tuned together with an fp32 delta vector — one fused
    optimizer per dtype tied by a joint clip coefficient."""
    joint = False
    if per_param_clip:
        groups = [[p] for p in params]
    elif len({p.dtype for p in params}) > 1:
        groups = [[p for p in params if p.dtype == dt] for dt in (torch.bfloat16, torch.float32)]
        joint = True
    else:
        groups = [params]
    opts = [FusedAdamWClip(g, lr=lr, betas=(0.9, 0.999), weight_decay=0.01, eps=1e-15) for g in groups]

    def clip_and_step():
        live = [o for o in opts if any(p.grad is not None for p in o.params)]
        if joint:
            FusedAdamWClip.joint_clip_grad_norm_(opts, 1.0)
        for o in live:
            if not joint:
                o.clip_grad_norm_(1.0)
            o.step()

    feed = _OneVideo(cond_latents, train_latents, prompt_embeds, prompt_mask, train_latents_variants)
    out = run_adaptation(wrapper, params, opts, _fm_loss(wrapper, feed, device, dtype), clip_and_step, num_steps,
                         early_stopper=early_stopper, finish_eval=False)
    return out["losses"], out["es_check_time"], out["early_stopping_info"]


def optimize_delta_a(wrapper: DeltaAWrapper, cond_latents, train_latents, prompt_embeds, prompt_mask, num_steps: int = 20,
                     lr: float = 1e-3, device: str = "cuda", dtype: torch.dtype = torch.bfloat16,
                     early_stopper: Optional[AnchoredEarlyStopper] = None,
                     train_latents_variants: Optional[List[Dict]] = None) -> Dict:
    losses, est, es_state = _optimize(wrapper, [wrapper.delta], False, cond_latents, train_latents, prompt_embeds,
                                      prompt_mask, num_steps, lr, device, dtype, early_stopper, train_latents_variants)
    return {"losses": losses, "delta_norm": wrapper.delta.detach().norm().item(), "es_check_time": est,
            "early_stopping_info": es_state}


def optimize_delta_b(wrapper: DeltaBWrapper, cond_latents, train_latents, prompt_embeds, prompt_mask, num_steps: int = 20,
                     lr: float = 1e-3, device: str = "cuda", dtype: torch.dtype = torch.bfloat16,
                     early_stopper: Optional[AnchoredEarlyStopper] = None,
                     train_latents_variants: Optional[List[Dict]] = None) -> Dict:
    params = list(wrapper.deltas) + ([wrapper.delta_final] if wrapper.delta_final is not None else [])
    losses, est, es_state = _optimize(wrapper, params, True, cond_latents, train_latents, prompt_embeds, prompt_mask,
                                      num_steps, lr, device, dtype, early_stopper, train_latents_variants)
    delta_norms = [d.detach().norm().item() for d in wrapper.deltas]
    if wrapper.delta_final is not None:   # run_delta_b.py:413-415
        delta_norms.append(wrapper.delta_final.detach().norm().item())
    return {"losses": losses, "delta_norms": delta_norms, "es_check_time": est, "early_stopping_info": es_state}


def optimize_delta_c(wrapper: DeltaCWrapper, cond_latents, train_latents, prompt_embeds, prompt_mask, num_steps: int = 20,
                     lr: float = 1e-3, device: str = "cuda", dtype: torch.dtype = torch.bfloat16,
                     early_stopper: Optional[AnchoredEarlyStopper] = None,
                     train_latents_variants: Optional[List[Dict]] = None) -> Dict:
    losses, est, es_state = _optimize(wrapper, [wrapper.delta_out], False, cond_latents, train_latents, prompt_embeds,
                                      prompt_mask, num_steps, lr, device, dtype, early_stopper, train_latents_variants)
    return {"losses": losses, "delta_out_norm": wrapper.delta_out.detach().norm().item(),           # run_delta_c.py:240-246
            "delta_out_values": wrapper.delta_out.detach().cpu().tolist(), "es_check_time": est, "early_stopping_info": es_state}


def optimize_film_adapter(wrapper: FiLMAdapterWrapper, cond_latents, train_latents, prompt_embeds, prompt_mask,
                          num_steps: int = 20, lr: float = 1e-3, device: str = "cuda", dtype: torch.dtype = torch.bfloat16,
                          early_stopper: Optional[AnchoredEarlyStopper] = None,
                          train_latents_variants: Optional[List[Dict]] = None) -> Dict:
    """run_film_tta.py:266-341: AdamW(eps 1e-15) over all corrections, one global clip at 1.0."""
    params = list(wrapper.corrections)
    losses, est, es_state = _optimize(wrapper, params, False, cond_latents, train_latents, prompt_embeds, prompt_mask,
                                      num_steps, lr, device, dtype, early_stopper, train_latents_variants)
    return {"losses": losses, "correction_norm": sum(c.detach().norm().item() for c in wrapper.corrections),
            "es_check_time": est, "early_stopping_info": es_state}


# --------------------------------------------------------------------------- norm tuning (run_norm_tune_tta.py)
def collect_norm_params(dit: nn.Module, norm_target: str) -> List[nn.Parameter]:
    """The affine norm parameters the reference unfreezes (run_norm_tune_tta.py:74-98): the cross-attention pre-norm
    (weight, bias) and / or the four q/k RMS-norm weights of every block, in the reference's order."""
    if norm_target not in ("cross_attn_norm", "qk_norm", "all_norm"):
        raise ValueError(f"Unknown norm_target: {norm_target}")
    params = []
    for blk in dit.blocks:
        if norm_target in ("cross_attn_norm", "all_norm"):
            n = blk.pre_crs_attn_norm
            if getattr(n, "weight", None) is not None:
                params.append(n.weight)
            if getattr(n, "bias", None) is not None:
                params.append(n.bias)
        if norm_target in ("qk_norm", "all_norm"):
            for mod in (blk.attn.q_norm, blk.attn.k_norm, blk.cross_attn.q_norm, blk.cross_attn.k_norm):
                if getattr(mod, "weight", None) is not None:
                    params.append(mod.weight)
    return params


class NormTuneForward(nn.Module):
    """Thin wrapper: the DiT itself carries the unfrozen norm parameters (run_norm_tune_tta.py:116-208); `apply_to_dit` /
    `remove_from_dit` exist so the shared runner loop can treat it like the hook-based wrappers (the tuned weights are in
    the modules already; `restore()` puts the per-job originals back for the next video)."""

    def __init__(self, dit: nn.Module, norm_target: Optional[str] = None, also_tune_delta: bool = False):
        """`NormTuneForward(dit)` is the reference's constructor (run_norm_tune_tta.py:122-124: the parameter list lives in its
        main()); with `norm_target` this wrapper does main()'s set-up too (:371-393: freeze, collect + unfreeze, optional
        delta-A vector) and keeps the originals for the per-video reset."""
        super().__init__()
        self.dit = dit
        for p in dit.parameters():
            p.requires_grad = False
        self.norm_params = collect_norm_params(dit, norm_target) if norm_target is not None else []
        for p in self.norm_params:
            p.requires_grad = True
        self._orig = [p.data.clone() for p in self.norm_params]
        # --also-tune-delta (run_norm_tune_tta.py:380-391): a delta-A vector (fp32, like the reference's
        # nn.Parameter(torch.zeros(adaln_dim))) added to the t_embedder output by a forward hook and tuned in the SAME
        # optimizer; the hook stays for the video's continuation and goes away in restore()
        self.delta = None
        self._delta_hook = None
        if also_tune_delta:
            dev = next(dit.parameters()).device
            self.delta = nn.Parameter(torch.zeros(dit.config.adaln_tembed_dim, device=dev))
            delta = self.delta
            self._delta_hook = dit.t_embedder.register_forward_hook(lambda _m, _i, out: out + delta.unsqueeze(0).to(out.dtype))

    @property
    def tuned_params(self):
        return self.norm_params + ([self.delta] if self.delta is not None else [])

    @property
    def config(self):
        return self.dit.config

    def apply_to_dit(self):
        pass

    def remove_from_dit(self):
        pass

    def restore(self):
        for p, o in zip(self.norm_params, self._orig):
            p.data.copy_(o)
            p.requires_grad = False
            p.grad = None
        if self._delta_hook is not None:
            self._delta_hook.remove()
            self._delta_hook = None

    supports_cond_kv_cache = True

    def forward(self, hidden_states, timestep, encoder_hidden_states, encoder_attention_mask=None, num_cond_latents=0, **kw):
        cache_kw = {k: kw[k] for k in _HookedWrapper._CACHE_KW if k in kw}
        return self.dit(hidden_states=hidden_states, timestep=timestep, encoder_hidden_states=encoder_hidden_states,
                        encoder_attention_mask=encoder_attention_mask, num_cond_latents=num_cond_latents, **cache_kw)


def optimize_norm_params(wrapper: NormTuneForward, norm_params: List[nn.Parameter], cond_latents, train_latents, prompt_embeds,
                         prompt_mask, num_steps: int = 20, lr: float = 1e-3, device: str = "cuda",
                         dtype: torch.dtype = torch.bfloat16, early_stopper: Optional[AnchoredEarlyStopper] = None,
                         train_latents_variants: Optional[List[Dict]] = None) -> Dict:
    """run_norm_tune_tta.py:215-283 (same positional order: the parameter list is the second argument): AdamW(eps 1e-15) over
    `norm_params` (bf16 norm weights, plus the fp32 delta vector of --also-tune-delta), one global clip at 1.0.  Returns the
    reference's keys (`losses`, `early_stopping_info`) and, beside them, the check time and how far the parameters moved."""
    norm_params = list(norm_params)
    losses, est, es_state = _optimize(wrapper, norm_params, False, cond_latents, train_latents, prompt_embeds,
                                      prompt_mask, num_steps, lr, device, dtype, early_stopper, train_latents_variants)
    out = {"losses": losses, "early_stopping_info": es_state, "es_check_time": est}
    if getattr(wrapper, "_orig", None) and len(wrapper._orig) == len(wrapper.norm_params):
        out["norm_param_drift"] = sum((p.detach().float() - o.float()).norm().item() for p, o in zip(wrapper.norm_params, wrapper._orig))
    if getattr(wrapper, "delta", None) is not None:
        out["delta_norm"] = wrapper.delta.detach().norm().item()
    return out
