"""Temporal split of the conditioning-window latents and its frame-count mirrors (integer results: bit-exact).

Follows delta_experiment/scripts/common.py:1365-1401 (split), :1488-1517 (budget mirror), :589-593 (frame rounding).
"""
from typing import Dict, Optional, Tuple

import torch


def _split_sizes(t_total: int, num_context: int, holdout_fraction: float) -> Tuple[int, int, int]:
    """(cond, train, val) latent-frame counts of a clip of `t_total` latent frames: the context is capped so that at least one
    frame is left to train on; of the rest a `holdout_fraction` share (at least one frame) is held out for validation, unless
    that would leave nothing to train on — then everything non-context trains and nothing is held out.  The ONE place this
    arithmetic lives: the tensor split and the frame-budget mirror below both call it."""
    n_cond = min(int(num_context), t_total - 1)
    rest = t_total - n_cond
    n_val = max(1, int(rest * float(holdout_fraction)))
    if rest - n_val < 1:
        return n_cond, rest, 0
    return n_cond, rest - n_val, n_val


def split_tta_latents(latents: torch.Tensor, num_context_latents: int, holdout_fraction: float = 0.25
                      ) -> Tuple[torch.Tensor, torch.Tensor, Optional[torch.Tensor]]:
    """[B,C,T,H,W] -> (cond [.., T_cond, ..], train, val or None), contiguous slices along T."""
    n_cond, n_train, n_val = _split_sizes(latents.shape[2], num_context_latents, holdout_fraction)
    cond, train, val = torch.split(latents, [n_cond, n_train, latents.shape[2] - n_cond - n_train], dim=2)
    return cond.contiguous(), train.contiguous(), (val.contiguous() if n_val > 0 else None)


def _estimate_latent_len(num_pixel_frames: int, vae_t_scale: int = 4) -> int:
    n = max(1, int(num_pixel_frames))
    return 1 + (n - 1) // int(vae_t_scale)


def estimate_tta_split_budget(tta_total_frames: int, tta_context_frames: int, holdout_fraction: float = 0.25,
                              vae_t_scale: int = 4) -> Dict[str, int]:
    """The same split counted from PIXEL frame counts (what the CLI guards check before any video is encoded)."""
    t_total = _estimate_latent_len(tta_total_frames, vae_t_scale)
    n_cond, n_train, n_val = _split_sizes(t_total, _estimate_latent_len(tta_context_frames, vae_t_scale), holdout_fraction)
    return {"total_latents": t_total, "cond_latents": n_cond, "train_latents": n_train, "val_latents": n_val}


def num_frames_valid(num_frames: int, vae_temporal_factor: int = 4) -> int:
    """Round a requested frame count up to 4k + 1 (generate_video_continuation, common.py:589-593)."""
    return ((num_frames - 1 + vae_temporal_factor - 1) // vae_temporal_factor) * vae_temporal_factor + 1
