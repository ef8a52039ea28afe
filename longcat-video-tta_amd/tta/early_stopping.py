"""Anchored early stopping for the TTA inner loop.

Same class / argument / state names as delta_experiment/scripts/early_stopping.py:33-317: anchor loss at fixed sigmas x
fixed noise draws (seed = md5(video_id) based) on held-out latents, strict-improvement bookkeeping, `patience` and
`first_rise` strategies, in-memory best snapshot.
"""
import argparse
import hashlib
from typing import Callable, List, Optional, Tuple

import torch
import torch.nn as nn

from .flow_matching import compute_flow_matching_loss_conditioned_fixed


def add_early_stopping_args(parser: argparse.ArgumentParser):
    g = parser.add_argument_group("Early stopping")
    g.add_argument("--es-disable", action="store_true", default=False, help="Disable early stopping entirely.")
    g.add_argument("--es-check-every", type=int, default=5, help="Evaluate anchor loss every N training steps.")
    g.add_argument("--es-patience", type=int, default=3, help="Stop after this many checks without improvement.")
    g.add_argument("--es-anchor-sigmas", type=str, default="0.25,0.5,0.75",
                   help="Comma-separated sigma values for anchor loss.")
    g.add_argument("--es-noise-draws", type=int, default=2, help="Number of noise draws per anchor sigma.")
    g.add_argument("--es-strategy", type=str, default="patience", choices=["patience", "first_rise"],
                   help="Stopping strategy.")
    g.add_argument("--es-holdout-fraction", type=float, default=0.25,
                   help="Fraction of non-context conditioning frames held out for anchor loss.")


def build_early_stopper_from_args(args) -> Optional["AnchoredEarlyStopper"]:
    if getattr(args, "es_disable", False):
        return None
    anchor_sigmas = [float(x) for x in args.es_anchor_sigmas.split(",")]
    return AnchoredEarlyStopper(check_every=args.es_check_every, patience=args.es_patience,
                                anchor_sigmas=anchor_sigmas, noise_draws=args.es_noise_draws,
                                strategy=args.es_strategy)


def es_seed_base(video_id: str) -> int:
    return int(hashlib.md5(video_id.encode()).hexdigest()[:8], 16) % (2 ** 31)


class AnchoredEarlyStopper:
    def __init__(self, check_every: int = 5, patience: int = 3, anchor_sigmas: Optional[List[float]] = None,
                 noise_draws: int = 2, strategy: str = "patience"):
        self.check_every = check_every
        self.patience = patience
        self.anchor_sigmas = anchor_sigmas or [0.25, 0.5, 0.75]
        self.noise_draws = noise_draws
        self.strategy = strategy
        self._reset()

    def _reset(self):
        self.model = None
        self.cond_latents = self.val_latents = self.prompt_embeds = self.prompt_mask = None
        self.device = self.dtype = self.forward_fn = None
        self.fixed_noises: List[torch.Tensor] = []
        self.best_loss = float("inf")
        self.best_state = None
        self.checks_without_improvement = 0
        self.step_count = 0
        self.stopped_early = False
        self.best_step = 0
        self.loss_history: List[Tuple[int, float]] = []

    def setup(self, model: nn.Module, cond_latents, val_latents, prompt_embeds, prompt_mask, device: str = "cuda",
              dtype: torch.dtype = torch.bfloat16, forward_fn: Optional[Callable] = None, video_id: str = "",
              save_fn: Optional[Callable] = None):
        self._reset()
        self.model, self.cond_latents, self.val_latents = model, cond_latents, val_latents
        self.prompt_embeds, self.prompt_mask = prompt_embeds, prompt_mask
        self.device, self.dtype, self.forward_fn = device, dtype, forward_fn
        seed_base = es_seed_base(video_id)
        self.fixed_noises = []
        for draw_idx in range(self.noise_draws):
            gen = torch.Generator(device=device)
            gen.manual_seed(seed_base + draw_idx)
            self.fixed_noises.append(torch.randn(val_latents.shape, generator=gen, device=device,
                                                 dtype=val_latents.dtype))
        self.best_state = save_fn() if save_fn is not None else self._default_snapshot()
        self.best_loss = self._compute_anchor_loss()
        self.loss_history.append((0, self.best_loss))

    def step(self, current_step: int, save_fn: Optional[Callable] = None) -> Tuple[bool, dict]:
        self.step_count = current_step
        if current_step == 0 or current_step % self.check_every != 0:
            return False, {}
        loss = self._compute_anchor_loss()
        self.loss_history.append((current_step, loss))
        improved = loss < self.best_loss
        if improved:
            self.best_loss = loss
            self.best_step = current_step
            self.best_state = save_fn() if save_fn is not None else self._default_snapshot()
            self.checks_without_improvement = 0
        else:
            self.checks_without_improvement += 1
        info = {"anchor_loss": loss, "best_loss": self.best_loss, "best_step": self.best_step,
                "checks_without_improvement": self.checks_without_improvement}
        should_stop = False
        if self.strategy == "patience":
            should_stop = self.checks_without_improvement >= self.patience
        elif self.strategy == "first_rise":
            should_stop = not improved and current_step > 0
        if should_stop:
            self.stopped_early = True
        return should_stop, info

    def restore(self, restore_fn: Optional[Callable] = None):
        if self.best_state is None:
            return
        if restore_fn is not None:
            restore_fn(self.best_state)
        elif self.model is not None:
            self.model.load_state_dict(self.best_state, strict=False)

    @property
    def state(self) -> Optional[dict]:
        if not self.loss_history:
            return None
        return {"stopped_early": self.stopped_early, "best_step": self.best_step, "best_loss": self.best_loss,
                "total_checks": len(self.loss_history), "loss_history": self.loss_history}

    def _default_snapshot(self) -> dict:
        if self.model is None:
            return {}
        trainable = {n for n, p in self.model.named_parameters() if p.requires_grad}
        return {k: v.detach().clone() for k, v in self.model.state_dict().items() if v.requires_grad or k in trainable}

    def _compute_anchor_loss(self) -> float:
        if self.val_latents is None or self.model is None:
            return float("inf")
        was_training = self.model.training
        self.model.eval()
        loss = compute_flow_matching_loss_conditioned_fixed(
            dit=self.model, cond_latents=self.cond_latents, target_latents=self.val_latents,
            prompt_embeds=self.prompt_embeds, prompt_mask=self.prompt_mask, fixed_sigmas=self.anchor_sigmas,
            fixed_noises=self.fixed_noises, device=self.device, dtype=self.dtype, forward_fn=self.forward_fn)
        if was_training:
            self.model.train()
        return loss
