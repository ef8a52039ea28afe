"""Full-model TTA: every DiT parameter trainable (lora_experiment/scripts/run_full_tta.py:95-228).

`finetune_full_on_conditioning` keeps the reference's loop — zero_grad, linear LR warm-up, augmentation-variant draw,
conditioning-aware flow-matching loss, backward, `clip_grad_norm_`, SGD(momentum 0) or AdamW step, early stopping on the
stopper's own snapshot of the trainable state — on the fused multi-tensor optimizers (`FusedSGDClip` / `FusedAdamWClip`:
two launches over the ~700 parameter tensors instead of ~10 foreach launches over each).  Gradients of the dense weights
come from `lcv_transpose_pad` + the NT GEMM over the token axis, see lcv_hip/autograd_ops.py.

`snapshot_base_state` / `reset_dit_weights` are the per-video reset (:222-228, :462); the reference parks the base copy
in host memory because an 80-141 GB GPU cannot hold a second model — 288 GB of HBM can, so the copy stays on the device
and the reset is a device-to-device copy.
"""
import time
from typing import Dict, List, Optional

import torch
import torch.nn as nn

from lcv_hip.ops import FusedAdamWClip, FusedSGDClip
from .early_stopping import AnchoredEarlyStopper
from .flow_matching import compute_flow_matching_loss_conditioned


def snapshot_base_state(dit: nn.Module) -> Dict[str, torch.Tensor]:
    return {k: v.detach().clone() for k, v in dit.state_dict().items()}


def reset_dit_weights(dit: nn.Module, base_state: Dict[str, torch.Tensor]) -> None:
    """run_full_tta.py:222-228: copy the base values back into every named parameter."""
    with torch.no_grad():
        for name, param in dit.named_parameters():
            if name in base_state:
                param.copy_(base_state[name].to(param.device))
            param.grad = None


def finetune_full_on_conditioning(dit: nn.Module, cond_latents: torch.Tensor, train_latents: torch.Tensor,
                                  prompt_embeds: torch.Tensor, prompt_mask: torch.Tensor, num_steps: int = 10,
                                  lr: float = 1e-5, warmup_steps: int = 2, weight_decay: float = 0.01,
                                  max_grad_norm: float = 1.0, device: str = "cuda", dtype: torch.dtype = torch.bfloat16,
                                  early_stopper: Optional[AnchoredEarlyStopper] = None,
                                  train_latents_variants: Optional[List[Dict]] = None, optimizer_type: str = "sgd") -> Dict:
    params = [p for p in dit.parameters() if p.requires_grad]
    if not params:
        raise ValueError("No trainable parameters found. Did you unfreeze the model?")
    if optimizer_type == "adamw":
        optimizer = FusedAdamWClip(params, lr=lr, betas=(0.9, 0.999), weight_decay=weight_decay, eps=1e-8)
    else:
        optimizer = FusedSGDClip(params, lr=lr, weight_decay=weight_decay)
    if train_latents_variants is None:
        train_latents_variants = [{"latents": train_latents, "name": "orig"}]
    dit.train()
    losses = []
    train_start = time.time()
    es_check_time = 0.0
    for step in range(num_steps):
        optimizer.zero_grad(set_to_none=True)
        if step < warmup_steps and warmup_steps > 0:
            for pg in optimizer.param_groups:
                pg["lr"] = lr * (step + 1) / warmup_steps
        vi = torch.randint(0, len(train_latents_variants), (1,)).item()
        loss = compute_flow_matching_loss_conditioned(dit=dit, cond_latents=cond_latents,
                                                      target_latents=train_latents_variants[vi]["latents"],
                                                      prompt_embeds=prompt_embeds, prompt_mask=prompt_mask, device=device,
                                                      dtype=dtype)
        loss.backward()
        optimizer.clip_grad_norm_(max_grad_norm)
        optimizer.step()
        losses.append(loss.item())
        del loss
        if early_stopper is not None:
            t0 = time.time()
            should_stop, es_info = early_stopper.step(step + 1)
            es_check_time += time.time() - t0
            if should_stop:
                print(f"  Early stopping at step {step + 1}: {es_info}")
                break
    torch.cuda.synchronize()
    train_time = time.time() - train_start
    dit.eval()
    es_state = None
    if early_stopper is not None:
        def _restore_full(state_dict):
            named = dict(dit.named_parameters())
            with torch.no_grad():
                for k, v in state_dict.items():
                    if k in named:
                        named[k].copy_(v)
        early_stopper.restore(restore_fn=_restore_full)
        es_state = early_stopper.state
    optimizer.zero_grad(set_to_none=True)          # 27 GB of gradients are not needed during the continuation
    return {"losses": losses, "train_time": train_time, "es_check_time": es_check_time, "early_stopping_info": es_state}


def finetune_full_batch(dit: nn.Module, batch_data: List[Dict], num_steps: int = 10, lr: float = 1e-5, warmup_steps: int = 2,
                        weight_decay: float = 0.01, max_grad_norm: float = 1.0, device: str = "cuda",
                        dtype: torch.dtype = torch.bfloat16, optimizer_type: str = "sgd") -> Dict:
    """run_full_tta.py:230-306: all parameters trained round-robin over several videos (step k takes video k % n; the
    tensors of a video are moved to the device when its turn comes), no early stopping."""
    params = [p for p in dit.parameters() if p.requires_grad]
    if not params:
        raise ValueError("No trainable parameters found. Did you unfreeze the model?")
    if optimizer_type == "adamw":
        optimizer = FusedAdamWClip(params, lr=lr, betas=(0.9, 0.999), weight_decay=weight_decay, eps=1e-8)
    else:
        optimizer = FusedSGDClip(params, lr=lr, weight_decay=weight_decay)
    dit.train()
    losses = []
    n_vids = len(batch_data)
    train_start = time.time()
    for step in range(num_steps):
        optimizer.zero_grad(set_to_none=True)
        if step < warmup_steps and warmup_steps > 0:
            for pg in optimizer.param_groups:
                pg["lr"] = lr * (step + 1) / warmup_steps
        bd = batch_data[step % n_vids]
        pm = bd["prompt_mask"].to(device) if bd["prompt_mask"] is not None else None
        loss = compute_flow_matching_loss_conditioned(dit=dit, cond_latents=bd["cond_latents"].to(device),
                                                      target_latents=bd["train_latents"].to(device),
                                                      prompt_embeds=bd["prompt_embeds"].to(device), prompt_mask=pm,
                                                      device=device, dtype=dtype)
        loss.backward()
        optimizer.clip_grad_norm_(max_grad_norm)
        optimizer.step()
        losses.append(loss.item())
        del loss
    torch.cuda.synchronize()
    train_time = time.time() - train_start
    dit.eval()
    optimizer.zero_grad(set_to_none=True)
    return {"losses": losses, "train_time": train_time, "es_check_time": 0.0, "early_stopping_info": None}
