"""The TTA inner loop: forward -> LoRA-only backward -> fused clip + AdamW, per test video.

Mirrors `finetune_lora_on_conditioning` / `finetune_lora_batch` of lora_experiment/scripts/run_lora_tta.py:425-634
(arguments, warm-up rule `lr*(step+1)/warmup` for step < warmup, random augmentation-variant pick, early-stopper
hook, returned dict keys).  The optimizer is the two-launch fused clip + AdamW (lcv_hip.ops.FusedAdamWClip).
"""
import time
from typing import Dict, List, Optional

import torch
import torch.nn as nn

from lcv_hip.ops import FusedAdamWClip

from .early_stopping import AnchoredEarlyStopper
from .flow_matching import compute_flow_matching_loss_conditioned
from .lora import get_lora_parameters


def choose_gradient_checkpointing(dit: nn.Module, num_tokens: int, mode: str = None) -> bool:
    """The reference always checkpoints every block (lora_experiment/scripts/run_lora_tta.py:806-811) because an 80-141 GB
    GPU has to; an MI355X has 288 GB.  `mode` (default: env LCV_TTA_CHECKPOINT, else "auto"): "on" / "off" / "auto".
    auto keeps activations resident (no second forward, measured -23% step time at 25 200 tokens) when the estimate
    tokens x depth x 150 KB (measured: 3.7 GB per block at 25 200 tokens, C = 4096) plus what is already allocated stays
    under 90% of the device memory."""
    import os
    from functools import partial
    from torch.utils.checkpoint import checkpoint
    mode = (mode or os.environ.get("LCV_TTA_CHECKPOINT", "auto")).lower()
    if mode == "auto":
        dev = next(dit.parameters()).device
        total = torch.cuda.get_device_properties(dev).total_memory
        scale = dit.config.hidden_size / 4096.0
        need = num_tokens * len(dit.blocks) * 150e3 * scale + torch.cuda.memory_allocated(dev) + 30e9 * scale * scale
        use = need > 0.90 * total
    else:
        use = mode != "off"
    dit.gradient_checkpointing = bool(use)
    dit._gradient_checkpointing_func = partial(checkpoint, use_reentrant=False) if use else None
    return bool(use)


def _restore_lora_from_state(model: nn.Module, state_dict: dict):
    current = model.state_dict()
    for k, v in state_dict.items():
        if k in current:
            current[k].copy_(v)


def finetune_lora_on_conditioning(dit: nn.Module, lora_modules, cond_latents: torch.Tensor,
                                  train_latents: torch.Tensor, prompt_embeds: torch.Tensor,
                                  prompt_mask: torch.Tensor, num_steps: int = 20, lr: float = 2e-4,
                                  warmup_steps: int = 3, weight_decay: float = 0.01, max_grad_norm: float = 1.0,
                                  device: str = "cuda", dtype: torch.dtype = torch.bfloat16,
                                  early_stopper: Optional[AnchoredEarlyStopper] = None, lora_param_fn=None,
                                  train_latents_variants: Optional[List[Dict]] = None) -> Dict:
    lora_params = lora_param_fn() if lora_param_fn is not None else get_lora_parameters(lora_modules)
    if not lora_params:
        raise ValueError("No LoRA parameters found.")
    optimizer = FusedAdamWClip(lora_params, lr=lr, betas=(0.9, 0.999), weight_decay=weight_decay, eps=1e-8)
    if train_latents_variants is None:
        train_latents_variants = [{"latents": train_latents, "name": "orig"}]

    def _save_fn():
        return [p.data.clone() for p in lora_params]

    def _restore_from_snapshot(snapshot):
        if isinstance(snapshot, dict):
            _restore_lora_from_state(dit, snapshot)
            return
        for p, saved in zip(lora_params, snapshot):
            p.data.copy_(saved)

    dit.train()
    losses = []
    train_start = time.time()
    es_check_time = 0.0
    for step in range(num_steps):
        optimizer.zero_grad(set_to_none=True)
        if step < warmup_steps and warmup_steps > 0:
            warmup_lr = lr * (step + 1) / warmup_steps
            for pg in optimizer.param_groups:
                pg["lr"] = warmup_lr
        vi = torch.randint(0, len(train_latents_variants), (1,)).item()
        step_train = train_latents_variants[vi]["latents"]
        loss = compute_flow_matching_loss_conditioned(dit=dit, cond_latents=cond_latents, target_latents=step_train,
                                                      prompt_embeds=prompt_embeds, prompt_mask=prompt_mask,
                                                      device=device, dtype=dtype)
        loss.backward()
        if getattr(dit, "_sp_group", None) is not None:   # frame-sharded forward: the adapter gradients are partial sums
            dit.sequence_parallel_sync_grads(optimizer.params)
        optimizer.clip_grad_norm_(max_grad_norm)
        optimizer.step()
        losses.append(loss.item())
        del loss
        if early_stopper is not None:
            es_t0 = time.time()
            should_stop, es_info = early_stopper.step(step + 1, save_fn=_save_fn)
            es_check_time += time.time() - es_t0
            if should_stop:
                print(f"  Early stopping at step {step + 1}: {es_info}")
                break
    train_time = time.time() - train_start
    dit.eval()
    es_state = None
    if early_stopper is not None:
        early_stopper.restore(restore_fn=_restore_from_snapshot)
        es_state = early_stopper.state
    return {"losses": losses, "train_time": train_time, "es_check_time": es_check_time,
            "early_stopping_info": es_state}


def finetune_lora_batch(dit: nn.Module, lora_modules, batch_data: List[Dict], num_steps: int = 20, lr: float = 2e-4,
                        warmup_steps: int = 3, weight_decay: float = 0.01, max_grad_norm: float = 1.0,
                        device: str = "cuda", dtype: torch.dtype = torch.bfloat16, lora_param_fn=None) -> Dict:
    """Shared adapters trained round-robin over several videos (retrieval-augmented batch TTA); no early stopping."""
    lora_params = lora_param_fn() if lora_param_fn is not None else get_lora_parameters(lora_modules)
    optimizer = FusedAdamWClip(lora_params, lr=lr, betas=(0.9, 0.999), weight_decay=weight_decay, eps=1e-8)
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
    train_time = time.time() - train_start
    dit.eval()
    return {"losses": losses, "train_time": train_time, "es_check_time": 0.0, "early_stopping_info": None}
