"""Conditioned rectified-flow loss of the TTA inner loop on HIP kernels.

Mirrors delta_experiment/scripts/common.py:414-489 (`compute_flow_matching_loss_conditioned`) and :492-559
(`..._fixed`): only the target latents are noised, `[cond_clean | noisy_target]` is concatenated on T, the per-frame
timestep is 0 on conditioning frames and sigma*1000 (round-tripped through bf16) on target frames, the DiT is called
with `num_cond_latents`, and the fp32 MSE is taken on the target slice against (eps - x0).
The noise mix and the MSE (+ its gradient) are single HIP kernels; sigma / eps sampling stays on the torch generator.
"""
from typing import List, Optional

import torch

from lcv_hip import ops


def _get_model_config(dit):
    if hasattr(dit, "config"):
        return dit.config
    if hasattr(dit, "dit") and hasattr(dit.dit, "config"):
        return dit.dit.config
    raise AttributeError(f"Cannot find config on {type(dit).__name__}. "
                         "Ensure the model or its .dit attribute inherits from ConfigMixin.")


class _FMMse(torch.autograd.Function):
    @staticmethod
    def forward(ctx, pred, eps, x0, Tc):
        loss, dpred = ops.fm_mse(pred, eps, x0, Tc, need_grad=True)
        ctx.save_for_backward(dpred)
        return loss.clone()

    @staticmethod
    def backward(ctx, g):
        (dpred,) = ctx.saved_tensors
        return dpred * g, None, None, None


def fm_mse_loss(pred: torch.Tensor, noise: torch.Tensor, target: torch.Tensor, T_cond: int) -> torch.Tensor:
    """mean((pred[:, :, T_cond:] - (noise - target))^2) in fp32; differentiable w.r.t. pred."""
    pred = pred.to(torch.float32).contiguous()
    # the kernel evaluates the reference's `(noise - target)` in bf16 (common.py:486 on bf16 latents); fp32 latents
    # (an fp32 VAE, pre-encoded blobs saved in fp32) are brought to that dtype here rather than reinterpreted
    noise, target = noise.to(torch.bfloat16), target.to(torch.bfloat16)
    if torch.is_grad_enabled() and pred.requires_grad:
        return _FMMse.apply(pred, noise, target, T_cond)
    loss, _ = ops.fm_mse(pred, noise, target, T_cond, need_grad=False)
    return loss


def _build_inputs(cond_latents, target_latents, sigma, noise, patch_t, num_train_timesteps, dtype, device):
    B, C, T_cond, H, W = cond_latents.shape
    T_target = target_latents.shape[2]
    N_cond, N_target = T_cond // patch_t, T_target // patch_t
    N_total = (T_cond + T_target) // patch_t
    noisy_target = ops.fm_noise(target_latents.to(torch.bfloat16), noise.to(torch.bfloat16),
                                sigma.to(torch.float32).expand(B).contiguous())
    hidden_states = torch.cat([cond_latents.to(dtype), noisy_target.to(dtype)], dim=2)
    timestep = torch.zeros(B, N_total, device=device, dtype=dtype)
    timestep[:, N_cond:] = (sigma * num_train_timesteps).unsqueeze(1).expand(B, N_target).to(dtype)
    return hidden_states, timestep, N_cond


def compute_flow_matching_loss_conditioned(dit, cond_latents, target_latents, prompt_embeds, prompt_mask,
                                           num_train_timesteps: int = 1000, sigma_min: float = 0.001,
                                           sigma_max: float = 1.0, device: str = "cuda",
                                           dtype: torch.dtype = torch.bfloat16, forward_fn=None) -> torch.Tensor:
    cfg = _get_model_config(dit)
    B = cond_latents.shape[0]
    T_cond = cond_latents.shape[2]
    sigma = torch.rand(B, device=device, dtype=torch.float32) * (sigma_max - sigma_min) + sigma_min
    noise = torch.randn_like(target_latents)
    hidden_states, timestep, N_cond = _build_inputs(cond_latents, target_latents, sigma, noise, cfg.patch_size[0],
                                                    num_train_timesteps, dtype, device)
    if forward_fn is not None:
        pred = forward_fn(hidden_states, timestep, N_cond)
    else:
        pred = dit(hidden_states=hidden_states, timestep=timestep, encoder_hidden_states=prompt_embeds,
                   encoder_attention_mask=prompt_mask, num_cond_latents=N_cond)
    return fm_mse_loss(pred, noise, target_latents, T_cond)


def compute_flow_matching_loss_conditioned_fixed(dit, cond_latents, target_latents, prompt_embeds, prompt_mask,
                                                 fixed_sigmas: List[float], fixed_noises: List[torch.Tensor],
                                                 num_train_timesteps: int = 1000, device: str = "cuda",
                                                 dtype: torch.dtype = torch.bfloat16, forward_fn=None) -> float:
    """Anchor loss of the early stopper: mean over sigmas x noise draws of the no-grad conditioned loss."""
    cfg = _get_model_config(dit)
    T_cond = cond_latents.shape[2]
    total, count = 0.0, 0
    for sigma_val in fixed_sigmas:
        sigma = torch.tensor([sigma_val], device=device, dtype=torch.float32)
        for noise in fixed_noises:
            hidden_states, timestep, N_cond = _build_inputs(cond_latents, target_latents, sigma, noise,
                                                            cfg.patch_size[0], num_train_timesteps, dtype, device)
            with torch.no_grad():
                if forward_fn is not None:
                    pred = forward_fn(hidden_states, timestep, N_cond)
                else:
                    pred = dit(hidden_states=hidden_states, timestep=timestep, encoder_hidden_states=prompt_embeds,
                               encoder_attention_mask=prompt_mask, num_cond_latents=N_cond)
                total += fm_mse_loss(pred, noise, target_latents, T_cond).item()
            count += 1
    return total / max(count, 1)


# ---- the unconditioned variants (common.py:274-343, 346-407).  Every reference runner imports them and none calls them
# (SURVEY §8 row a3); they are kept for API parity on the same two kernels (lcv_fm_noise, lcv_fm_mse with T_cond = 0).
def compute_flow_matching_loss(dit, latents, prompt_embeds, prompt_mask, num_train_timesteps: int = 1000,
                               sigma_min: float = 0.001, sigma_max: float = 1.0, device: str = "cuda",
                               dtype: torch.dtype = torch.bfloat16, forward_fn=None) -> torch.Tensor:
    cfg = _get_model_config(dit)
    B, T_lat = latents.shape[0], latents.shape[2]
    sigma = torch.rand(B, device=device, dtype=torch.float32) * (sigma_max - sigma_min) + sigma_min
    noise = torch.randn_like(latents)
    noisy = ops.fm_noise(latents.to(torch.bfloat16), noise.to(torch.bfloat16), sigma.to(torch.float32).expand(B).contiguous()).to(dtype)
    timestep = (sigma * num_train_timesteps).unsqueeze(1).expand(B, T_lat // cfg.patch_size[0]).to(dtype)
    if forward_fn is not None:
        pred = forward_fn(noisy, timestep)
    else:
        pred = dit(hidden_states=noisy, timestep=timestep, encoder_hidden_states=prompt_embeds, encoder_attention_mask=prompt_mask)
    return fm_mse_loss(pred, noise, latents, 0)


def compute_flow_matching_loss_fixed(dit, latents, prompt_embeds, prompt_mask, fixed_sigmas: List[float], noise_draws: int = 1,
                                     num_train_timesteps: int = 1000, device: str = "cuda",
                                     dtype: torch.dtype = torch.bfloat16, forward_fn=None) -> float:
    cfg = _get_model_config(dit)
    B, T_lat = latents.shape[0], latents.shape[2]
    total, count = 0.0, 0
    for sigma_val in fixed_sigmas:
        sigma = torch.tensor([sigma_val], device=device, dtype=torch.float32)
        timestep = (sigma * num_train_timesteps).unsqueeze(1).expand(B, T_lat // cfg.patch_size[0]).to(dtype)
        for draw_idx in range(noise_draws):
            gen = torch.Generator(device=device)
            gen.manual_seed(42 + draw_idx)                       # the reference's fixed seeds (:385-386)
            noise = torch.randn(latents.shape, generator=gen, device=device, dtype=latents.dtype)
            noisy = ops.fm_noise(latents.to(torch.bfloat16), noise.to(torch.bfloat16), sigma.expand(B).contiguous()).to(dtype)
            with torch.no_grad():
                if forward_fn is not None:
                    pred = forward_fn(noisy, timestep)
                else:
                    pred = dit(hidden_states=noisy, timestep=timestep, encoder_hidden_states=prompt_embeds,
                               encoder_attention_mask=prompt_mask)
                total += fm_mse_loss(pred, noise, latents, 0).item()
            count += 1
    return total / max(count, 1)
