"""Temporal split of the conditioning-window latents and its frame-count mirrors (integer results: bit-exact).

Follows delta_experiment/scripts/common.py:1365-1401 (split), :1488-1517 (budget mirror), :589-593 (frame rounding).
"""
from typing import Dict, Optional, Tuple

import torch


def split_tta_latents(latents: torch.Tensor, num_context_latents: int, holdout_fraction: float = 0.25
                      ) -> Tuple[torch.Tensor, torch.Tensor, Optional[torch.Tensor]]:
    """[B,C,T,H,W] -> (cond [.., T_cond, ..], train, val or None).  At least one non-context frame is kept;
    when fewer than one train frame would remain everything non-context trains and val is None."""
    T_total = latents.shape[2]
    T_cond = min(num_context_latents, T_total - 1)
    remainder = T_total - T_cond
    T_val = max(1, int(remainder * holdout_fraction))
    T_train = remainder - T_val
    if T_train < 1:
        T_train = remainder
        T_val = 0
    cond = latents[:, :, :T_cond].contiguous()
    train = latents[:, :, T_cond:T_cond + T_train].contiguous()
    val = latents[:, :, T_cond + T_train:].contiguous() if T_val > 0 else None
    return cond, train, val


def _estimate_latent_len(num_pixel_frames: int, vae_t_scale: int = 4) -> int:
    n = max(1, int(num_pixel_frames))
    return 1 + (n - 1) // int(vae_t_scale)


def estimate_tta_split_budget(tta_total_frames: int, tta_context_frames: int, holdout_fraction: float = 0.25,
                              vae_t_scale: int = 4) -> Dict[str, int]:
    t_total = _estimate_latent_len(tta_total_frames, vae_t_scale)
    t_ctx_req = _estimate_latent_len(tta_context_frames, vae_t_scale)
    t_cond = min(t_ctx_req, t_total - 1)
    remainder = t_total - t_cond
    t_val = max(1, int(remainder * float(holdout_fraction)))
    t_train = remainder - t_val
    if t_train < 1:
        t_train = remainder
        t_val = 0
    return {"total_latents": int(t_total), "cond_latents": int(t_cond), "train_latents": int(t_train),
            "val_latents": int(t_val)}


def num_frames_valid(num_frames: int, vae_temporal_factor: int = 4) -> int:
    """Round a requested frame count up to 4k + 1 (generate_video_continuation, common.py:589-593)."""
    return ((num_frames - 1 + vae_temporal_factor - 1) // vae_temporal_factor) * vae_temporal_factor + 1
