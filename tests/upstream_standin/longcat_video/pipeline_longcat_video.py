"""Stand-in pipeline surface: `retrieve_latents`, the sigma grid, CFG-zero-star and the signed Euler step, as
spec/dit.md A11-A16 assume them (on oracle/pipeline_oracle.py)."""
import os
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[3]))
from oracle import pipeline_oracle as PO  # noqa: E402


def retrieve_latents(encoder_output, generator=None, sample_mode="sample"):
    d = encoder_output.latent_dist
    return d.sample(generator) if sample_mode == "sample" else d.mode()


class LongCatVideoPipeline:
    def __init__(self, tokenizer=None, text_encoder=None, vae=None, scheduler=None, dit=None):
        self.tokenizer, self.text_encoder, self.vae, self.scheduler, self.dit = tokenizer, text_encoder, vae, scheduler, dit

    def get_timesteps_sigmas(self, sampling_steps):
        lo = 0.01 if os.environ.get("STANDIN_BREAK") == "sigma_grid" else 0.001
        return torch.linspace(1, lo, sampling_steps, dtype=torch.float32)

    # the two arithmetic pieces of a denoise step, exposed so a guard can feed them fixed predictions
    def combine_cfg(self, cond, uncond, guidance):
        if os.environ.get("STANDIN_BREAK") == "plain_cfg":
            return uncond + guidance * (cond - uncond)
        return PO.cfg_zero_star(cond, uncond, guidance)

    def euler(self, x, v, dt):
        return PO.euler_update(x, v, dt, negate=os.environ.get("STANDIN_BREAK") != "sign")
