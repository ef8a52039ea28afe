"""The test-time-adaptation inner loop as ONE engine for every method of the path.

Contract kept from the reference: the public functions `finetune_lora_on_conditioning` / `finetune_lora_batch`
(lora_experiment/scripts/run_lora_tta.py:425-634) with their argument names and defaults, and
the returned keys `losses`, `train_time`, `es_check_time`, `early_stopping_info`.  Per step: gradients dropped, linear
warm-up `lr * (step + 1) / warmup` while `step < warmup` (the last warm-up step sets exactly `lr`, nothing resets it
afterwards), a uniformly drawn augmentation variant, the conditioned flow-matching loss, backward, global-norm clip,
optimizer step, early-stopper hook; at the end the stopper's best state is written back and the model leaves train mode.

The reference carries four copies of this loop (LoRA, full model, their batch forms) plus one per delta method.  Here
`run_adaptation` is the only loop; the public functions describe WHAT is trained and WHERE the data comes from:
  * the optimizer is the fused multi-tensor clip + AdamW / SGD (two launches per step, `lcv_hip.ops`);
  * per-step losses stay on the device in a preallocated log and come to the host in one copy — the host never waits
    for a step (the reference's `loss.item()` at :516 drains the stream every step); it synchronises only where the
    early stopper needs a number, and once at the end so `train_time` covers the queued work;
  * the early stopper's snapshot is one reusable buffer set (`ParamSnapshot`), not a clone per improvement;
  * batch TTA stages every video's tensors in HBM once (288 GB) instead of a host->device copy per step (:603-606).
"""
import os
import time
from functools import partial
from typing import Callable, Dict, List, Optional, Sequence

import torch
import torch.nn as nn
from torch.utils.checkpoint import checkpoint

from lcv_hip.ops import FusedAdamWClip

from .early_stopping import AnchoredEarlyStopper, ParamSnapshot
from .flow_matching import compute_flow_matching_loss_conditioned
from .lora import get_lora_parameters

_ACT_BYTES_PER_TOKEN_BLOCK = 150e3     # measured: 3.7 GB per block at 25 200 tokens, hidden 4096, no checkpointing


def choose_gradient_checkpointing(dit: nn.Module, num_tokens: int, mode: str = None) -> bool:
    """The reference always checkpoints every block (lora_experiment/scripts/run_lora_tta.py:806-811) because an 80-141 GB
    GPU has to; an MI355X has 288 GB.  `mode` (default: env LCV_TTA_CHECKPOINT, else "auto"): "on" / "off" / "auto".
    auto keeps activations resident (no second forward, measured -23% step time at 25 200 tokens) when the estimate
    tokens x depth x 150 KB plus what is already allocated stays under 90% of the device memory."""
    mode = (mode or os.environ.get("LCV_TTA_CHECKPOINT", "auto")).lower()
    if mode == "auto":
        dev = next(dit.parameters()).device
        width = dit.config.hidden_size / 4096.0
        need = (num_tokens * len(dit.blocks) * _ACT_BYTES_PER_TOKEN_BLOCK * width + torch.cuda.memory_allocated(dev)
                + 30e9 * width * width)
        use = need > 0.90 * torch.cuda.get_device_properties(dev).total_memory
    else:
        use = mode != "off"
    dit.gradient_checkpointing = bool(use)
    dit._gradient_checkpointing_func = partial(checkpoint, use_reentrant=False) if use else None
    return bool(use)


# ------------------------------------------------------------------------------------------------ the engine
class _LossLog:
    """Per-step losses parked on the device; one device->host copy when somebody needs the numbers."""

    def __init__(self, capacity: int, device):
        self.buf = torch.zeros(max(capacity, 1), dtype=torch.float32, device=device)
        self.count = 0

    def push(self, loss: torch.Tensor) -> None:
        self.buf[self.count].copy_(loss.detach().reshape(()), non_blocking=True)
        self.count += 1

    def to_list(self) -> List[float]:
        return self.buf[:self.count].tolist()


def _variant_pool(train_latents, variants) -> List[torch.Tensor]:
    if variants is None:
        return [train_latents]
    return [v["latents"] for v in variants]


def run_adaptation(module: nn.Module, params: Sequence[torch.Tensor], optimizers: Sequence, loss_at: Callable[[int], torch.Tensor],
                   clip_and_step: Callable[[], None], num_steps: int, lr: float = 0.0, warmup_steps: int = 0,
                   early_stopper: Optional[AnchoredEarlyStopper] = None, grad_sync: Optional[Callable[[], None]] = None,
                   finish_eval: bool = True) -> Dict:
    """`loss_at(step)` returns the differentiable loss of that step; `clip_and_step()` owns clipping + the update."""
    device = params[0].device
    log = _LossLog(num_steps, device)
    keeper = None
    if early_stopper is not None:
        # the stopper's own initial snapshot (taken in setup) is refreshed in place when it covers the same tensors
        held = early_stopper.best_state
        keeper = held if isinstance(held, ParamSnapshot) and held.covers(params) else ParamSnapshot(params)
    module.train()
    es_seconds = 0.0
    started = time.perf_counter()
    for step in range(num_steps):
        for opt in optimizers:
            opt.zero_grad(set_to_none=True)
        if 0 < warmup_steps and step < warmup_steps:
            for opt in optimizers:
                for group in opt.param_groups:
                    group["lr"] = lr * (step + 1) / warmup_steps
        loss = loss_at(step)
        loss.backward()
        if grad_sync is not None:
            grad_sync()
        clip_and_step()
        log.push(loss)
        del loss
        if early_stopper is None:
            continue
        done = step + 1
        due = done % early_stopper.check_every == 0
        if due:                       # the check's own time, not the wait for the training kernels queued before it
            torch.cuda.synchronize(device)
            tick = time.perf_counter()
        stop, info = early_stopper.step(done, save_fn=keeper.capture)
        if due:
            torch.cuda.synchronize(device)
            es_seconds += time.perf_counter() - tick
        if stop:
            print(f"  early stop after step {done}: {info}")
            break
    if early_stopper is not None:
        early_stopper.restore(restore_fn=_put_back(params))
    torch.cuda.synchronize(device)
    elapsed = time.perf_counter() - started
    if finish_eval:
        module.eval()
    return {"losses": log.to_list(), "train_time": elapsed, "es_check_time": es_seconds,
            "early_stopping_info": early_stopper.state if early_stopper is not None else None}


def _put_back(params: Sequence[torch.Tensor]) -> Callable:
    """Restore callback accepting every snapshot form the stopper may hold: the engine's `ParamSnapshot`, a caller's
    list of tensors in parameter order (the runners' `save_fn`), or a name -> tensor mapping (the stopper's fallback)."""
    def restore(snapshot):
        if isinstance(snapshot, ParamSnapshot):
            snapshot.write_back()
            return
        with torch.no_grad():
            if isinstance(snapshot, dict):
                raise TypeError("name-keyed snapshots must be restored by the model's owner (pass restore_fn)")
            torch._foreach_copy_([p.detach() for p in params], [s.to(p.device) for s, p in zip(snapshot, params)])
    return restore


# ------------------------------------------------------------------------------------------------ data feeds
class _OneVideo:
    def __init__(self, cond, train, embeds, mask, variants):
        self.cond, self.embeds, self.mask = cond, embeds, mask
        self.pool = _variant_pool(train, variants)

    def __call__(self, step: int):
        pick = torch.randint(0, len(self.pool), (1,)).item()      # host RNG draw every step, like the reference (:499)
        return self.cond, self.pool[pick], self.embeds, self.mask


class _RoundRobin:
    """Retrieval-augmented batch TTA: step k trains on video k % n (run_lora_tta.py:596-606).  All videos are staged
    in HBM up front — a few hundred MB against 288 GB — so no step waits on a host->device copy."""

    def __init__(self, batch_data: List[Dict], device):
        def dev(t):
            return None if t is None else t.to(device, non_blocking=True)
        self.items = [(dev(b["cond_latents"]), dev(b["train_latents"]), dev(b["prompt_embeds"]), dev(b.get("prompt_mask")))
                      for b in batch_data]

    def __call__(self, step: int):
        return self.items[step % len(self.items)]


def _fm_loss(dit, feed, device, dtype):
    def at(step: int) -> torch.Tensor:
        cond, target, embeds, mask = feed(step)
        return compute_flow_matching_loss_conditioned(dit=dit, cond_latents=cond, target_latents=target, prompt_embeds=embeds,
                                                      prompt_mask=mask, device=device, dtype=dtype)
    return at


def _single_optimizer_step(opt, max_grad_norm: float) -> Callable[[], None]:
    def go():
        opt.clip_grad_norm_(max_grad_norm)
        opt.step()
    return go


def _sp_sync(dit, opt) -> Optional[Callable[[], None]]:
    # frame-sharded forward (sequence parallelism): every rank holds partial adapter gradients
    if getattr(dit, "_sp_group", None) is None:
        return None
    return lambda: dit.sequence_parallel_sync_grads(opt.params)


# ------------------------------------------------------------------------------------------------ public: LoRA
def _adapter_params(lora_modules, lora_param_fn) -> List[torch.Tensor]:
    params = lora_param_fn() if lora_param_fn is not None else get_lora_parameters(lora_modules)
    if not params:
        raise ValueError("no LoRA parameters to train: inject adapters before the inner loop")
    return params


def finetune_lora_on_conditioning(dit: nn.Module, lora_modules, cond_latents: torch.Tensor,
                                  train_latents: torch.Tensor, prompt_embeds: torch.Tensor,
                                  prompt_mask: torch.Tensor, num_steps: int = 20, lr: float = 2e-4,
                                  warmup_steps: int = 3, weight_decay: float = 0.01, max_grad_norm: float = 1.0,
                                  device: str = "cuda", dtype: torch.dtype = torch.bfloat16,
                                  early_stopper: Optional[AnchoredEarlyStopper] = None, lora_param_fn=None,
                                  train_latents_variants: Optional[List[Dict]] = None) -> Dict:
    params = _adapter_params(lora_modules, lora_param_fn)
    opt = FusedAdamWClip(params, lr=lr, betas=(0.9, 0.999), weight_decay=weight_decay, eps=1e-8)
    feed = _OneVideo(cond_latents, train_latents, prompt_embeds, prompt_mask, train_latents_variants)
    return run_adaptation(dit, params, [opt], _fm_loss(dit, feed, device, dtype), _single_optimizer_step(opt, max_grad_norm),
                          num_steps, lr, warmup_steps, early_stopper, grad_sync=_sp_sync(dit, opt))


def finetune_lora_batch(dit: nn.Module, lora_modules, batch_data: List[Dict], num_steps: int = 20, lr: float = 2e-4,
                        warmup_steps: int = 3, weight_decay: float = 0.01, max_grad_norm: float = 1.0,
                        device: str = "cuda", dtype: torch.dtype = torch.bfloat16, lora_param_fn=None) -> Dict:
    """Shared adapters trained round-robin over the eval video and its neighbours; no early stopping (:558-634)."""
    params = _adapter_params(lora_modules, lora_param_fn)
    opt = FusedAdamWClip(params, lr=lr, betas=(0.9, 0.999), weight_decay=weight_decay, eps=1e-8)
    feed = _RoundRobin(batch_data, device)
    return run_adaptation(dit, params, [opt], _fm_loss(dit, feed, device, dtype), _single_optimizer_step(opt, max_grad_norm),
                          num_steps, lr, warmup_steps, None, grad_sync=_sp_sync(dit, opt))
