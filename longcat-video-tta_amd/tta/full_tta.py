"""Full-model TTA (`METHOD=full`): every DiT parameter is trained on the video's conditioning clip.

Contract kept from the reference (lora_experiment/scripts/run_full_tta.py:95-306, 444-462): the public
names `finetune_full_on_conditioning`, `finetune_full_batch`, `reset_dit_weights`, their arguments and defaults
(`optimizer_type` "sgd" = plain SGD with weight decay, momentum 0; "adamw" = AdamW(0.9, 0.999, eps 1e-8)), and the four
returned keys.  The loop itself is `inner_loop.run_adaptation` — the same engine the LoRA and delta methods use — with
  * the fused multi-tensor optimizers (`FusedSGDClip` / `FusedAdamWClip`: a norm launch and an update launch over the
    ~700 parameter tensors, torch's foreach rounding points);
  * the base copy for the per-video reset living in HBM (`snapshot_base_state`): the reference parks it in host memory
    because an 80-141 GB card cannot hold a second model (:459-462); 288 GB can, so the reset is one device-side
    multi-tensor copy;
  * the early stopper's best state held in one reusable 27 GB buffer set instead of a fresh state-dict clone per
    improvement, and the 27 GB of gradients released before the continuation starts.
Dense-weight gradients come from `lcv_transpose_pad` + the NT GEMM over the token axis (lcv_hip/autograd_ops.py).
"""
from typing import Dict, List, Optional

import torch
import torch.nn as nn

from lcv_hip.ops import FusedAdamWClip, FusedSGDClip

from .early_stopping import AnchoredEarlyStopper
from .inner_loop import _OneVideo, _RoundRobin, _fm_loss, _single_optimizer_step, run_adaptation


def snapshot_base_state(dit: nn.Module) -> Dict[str, torch.Tensor]:
    """Device-resident copy of every state-dict entry, taken once per job."""
    return {name: t.detach().clone() for name, t in dit.state_dict().items()}


def reset_dit_weights(dit: nn.Module, base_state: Dict[str, torch.Tensor]) -> None:
    """Per-video reset (run_full_tta.py:222-228): base values back into every named parameter, gradients dropped."""
    dst, src = [], []
    for name, p in dit.named_parameters():
        p.grad = None
        if name in base_state:
            dst.append(p.detach())
            src.append(base_state[name].to(p.device))
    if dst:
        with torch.no_grad():
            torch._foreach_copy_(dst, src)


def _trainable(dit: nn.Module) -> List[torch.Tensor]:
    params = [p for p in dit.parameters() if p.requires_grad]
    if not params:
        raise ValueError("nothing to train: unfreeze the DiT (requires_grad) before full-model TTA")
    return params


def _make_optimizer(kind: str, params, lr: float, weight_decay: float):
    if kind == "adamw":
        return FusedAdamWClip(params, lr=lr, betas=(0.9, 0.999), weight_decay=weight_decay, eps=1e-8)
    if kind == "sgd":
        return FusedSGDClip(params, lr=lr, weight_decay=weight_decay)
    raise ValueError(f"unknown optimizer_type {kind!r} (sgd | adamw)")


def _run(dit, feed, num_steps, lr, warmup_steps, weight_decay, max_grad_norm, device, dtype, early_stopper, optimizer_type):
    params = _trainable(dit)
    opt = _make_optimizer(optimizer_type, params, lr, weight_decay)
    out = run_adaptation(dit, params, [opt], _fm_loss(dit, feed, device, dtype), _single_optimizer_step(opt, max_grad_norm),
                         num_steps, lr, warmup_steps, early_stopper)
    opt.zero_grad(set_to_none=True)            # the gradients of 13.6 B parameters are dead weight during generation
    return out


def finetune_full_on_conditioning(dit: nn.Module, cond_latents: torch.Tensor, train_latents: torch.Tensor,
                                  prompt_embeds: torch.Tensor, prompt_mask: torch.Tensor, num_steps: int = 10,
                                  lr: float = 1e-5, warmup_steps: int = 2, weight_decay: float = 0.01,
                                  max_grad_norm: float = 1.0, device: str = "cuda", dtype: torch.dtype = torch.bfloat16,
                                  early_stopper: Optional[AnchoredEarlyStopper] = None,
                                  train_latents_variants: Optional[List[Dict]] = None, optimizer_type: str = "sgd") -> Dict:
    feed = _OneVideo(cond_latents, train_latents, prompt_embeds, prompt_mask, train_latents_variants)
    return _run(dit, feed, num_steps, lr, warmup_steps, weight_decay, max_grad_norm, device, dtype, early_stopper,
                optimizer_type)


def finetune_full_batch(dit: nn.Module, batch_data: List[Dict], num_steps: int = 10, lr: float = 1e-5, warmup_steps: int = 2,
                        weight_decay: float = 0.01, max_grad_norm: float = 1.0, device: str = "cuda",
                        dtype: torch.dtype = torch.bfloat16, optimizer_type: str = "sgd") -> Dict:
    """Round-robin over the eval video and its retrieved neighbours (run_full_tta.py:230-306); no early stopping."""
    return _run(dit, _RoundRobin(batch_data, device), num_steps, lr, warmup_steps, weight_decay, max_grad_norm, device, dtype,
                None, optimizer_type)
