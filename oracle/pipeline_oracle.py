"""CPU oracle for the N-step denoise loop behind `pipe.generate_vc` (TEST INFRASTRUCTURE — never imported by the
product; only `tests/`, `__graft_entry__.smoke()` and `bench.py::cpu_baseline` may use it).

What the reference does at this boundary (paths relative to /root/reference):
  * `generate_video_continuation` rounds the frame count, seeds a device generator and calls
    `pipe.generate_vc(video, prompt, resolution, num_frames, num_cond_frames, num_inference_steps, guidance_scale,
    generator, use_kv_cache=True, offload_kv_cache=False)[0]` ................ delta_experiment/scripts/common.py:566-611
  * the inference-only baseline makes the same call with `offload_kv_cache=True` ... baseline_experiment/scripts/run_baseline.py:409-420
  * the DiT call inside a step has the signature the loss uses ................ common.py:476-482; outer forward run_delta_a.py:134-217
Everything BELOW that call is the un-vendored `meituan-longcat/LongCat-Video` pipeline (unpinned HEAD,
PARTNER_SETUP_GUIDE.md:77-78), absent offline and without a fixture in the reference: **parity unpinned**.  The loop is
restated from the published algorithm; every item is listed with its guard test in `spec/dit.md`:
  [assumed-from-upstream] sigma grid `linspace(1, 0.001, n)`, static `shift` warp `s*sig / (1 + (s-1)*sig)`, trailing 0;
  [assumed-from-upstream] timestep fed to the DiT = sigma * 1000, cast to the model dtype (bf16 round trip);
  [assumed-from-upstream] CFG-zero-star: st = <c,u> / (<u,u> + 1e-8) per sample, v = u*st + g*(c - u*st);
  [assumed-from-upstream] `noise_pred = -noise_pred` before the scheduler step (the DiT predicts eps - x0 ... common.py:486);
  [assumed-from-upstream] Euler: x <- x + (sigma_next - sigma) * noise_pred, latents kept in fp32;
  [assumed-from-upstream] conditioning frames: either pinned in the sequence at timestep 0 and never updated, or — with
    `use_kv_cache` — passed ONCE through the DiT at t = 0 with cross-attention skipped, each block's (K, V) kept, and the
    steps run over the noise tokens only (`dit_oracle.dit_forward(..., kv_cache_dict=)`).
"""
from typing import Callable, Dict, List, Optional

import torch

from . import dit_oracle as D


def sigma_grid(num_steps: int, shift: float = 1.0, num_train_timesteps: int = 1000):
    """(timesteps[n], sigmas[n + 1]) of the flow-match Euler schedule on the pipeline's own grid."""
    sig = torch.linspace(1, 0.001, num_steps, dtype=torch.float32)
    sig = shift * sig / (1 + (shift - 1) * sig)
    return sig * num_train_timesteps, torch.cat([sig, torch.zeros(1)])


def cfg_zero_star(cond: torch.Tensor, uncond: torch.Tensor, guidance: float) -> torch.Tensor:
    """v = u*st + g*(c - u*st) with the per-sample projection st of the conditional on the unconditional prediction."""
    B = cond.shape[0]
    c, u = cond.reshape(B, -1).double(), uncond.reshape(B, -1).double()
    st = ((c * u).sum(1, keepdim=True) / ((u * u).sum(1, keepdim=True) + 1e-8)).float().view(B, *([1] * (cond.dim() - 1)))
    return uncond * st + guidance * (cond - uncond * st)


def euler_update(x: torch.Tensor, v: torch.Tensor, dt: float, negate: bool = True) -> torch.Tensor:
    return x + dt * (-v if negate else v)


def cache_clean_latents(P, cfg, cond_latents: torch.Tensor, bf16: bool = True) -> Dict[int, tuple]:
    """One pass over the clean conditioning latents at t = 0, text cross-attention skipped: per block (K pre-RoPE, V)."""
    B, _, Tc, _, _ = cond_latents.shape
    ts = torch.zeros(B, Tc, device=cond_latents.device)
    empty = torch.zeros(B, 1, 4, cfg["caption_channels"], device=cond_latents.device)
    _, kv = D.dit_forward(P, cfg, cond_latents, ts, empty, None, 0, bf16=bf16, return_kv=True, skip_crs_attn=True)
    return kv


def denoise(P: Dict[str, torch.Tensor], cfg: dict, latents: torch.Tensor, prompt_embeds, prompt_mask,
            negative_embeds=None, negative_mask=None, num_cond_latents: int = 0, num_inference_steps: int = 50,
            guidance_scale: float = 4.0, use_kv_cache: bool = True, shift: float = 1.0, bf16: bool = True,
            negate_pred: bool = True, zero_star: bool = True,
            step_callback: Optional[Callable[[int, torch.Tensor], None]] = None) -> torch.Tensor:
    """latents fp32 [1, C, T, h, w], the first `num_cond_latents` frames clean.  Returns the denoised latents (fp32);
    `step_callback(i, latents_after_step_i)` sees every intermediate state (full clip, cond frames included)."""
    rnd = D.bf16_round if bf16 else (lambda t: t)
    timesteps, sigmas = sigma_grid(num_inference_steps, shift)
    x = latents.float().clone()
    ncl = int(num_cond_latents)
    do_cfg = guidance_scale > 1.0 and negative_embeds is not None
    kv = None
    if ncl > 0 and use_kv_cache:
        cond = x[:, :, :ncl]
        kv = cache_clean_latents(P, cfg, rnd(cond), bf16)
        work = x[:, :, ncl:].clone()
    else:
        cond, work = None, x
    if do_cfg:
        emb = torch.cat([negative_embeds, prompt_embeds], 0)
        mask = None if prompt_mask is None else torch.cat([negative_mask, prompt_mask], 0)
    else:
        emb, mask = prompt_embeds, prompt_mask
    Bm, T_in = emb.shape[0], work.shape[2]
    for i in range(num_inference_steps):
        t = float(timesteps[i])
        x_in = rnd(work).expand(Bm, -1, -1, -1, -1)
        ts = rnd(torch.full((Bm, T_in), t, device=x.device))
        if kv is None and ncl > 0:
            ts[:, :ncl] = 0
        pred = D.dit_forward(P, cfg, x_in, ts, emb, mask, ncl, bf16=bf16, kv_cache_dict=kv)
        if kv is None and ncl > 0:
            pred = pred[:, :, ncl:]
        v = cfg_zero_star(pred[1:2], pred[0:1], guidance_scale) if (do_cfg and zero_star) else \
            (pred[0:1] + guidance_scale * (pred[1:2] - pred[0:1]) if do_cfg else pred)
        dt = float(sigmas[i + 1]) - float(sigmas[i])
        if kv is None and ncl > 0:
            work[:, :, ncl:] = euler_update(work[:, :, ncl:], v, dt, negate_pred)
        else:
            work = euler_update(work, v, dt, negate_pred)
        if step_callback is not None:
            step_callback(i, work if cond is None else torch.cat([cond, work], 2))
    return work if cond is None else torch.cat([cond, work], 2)
