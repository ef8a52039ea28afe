"""`LongCatVideoPipeline` — the denoise-and-decode driver behind the reference's
`generate_video_continuation` (delta_experiment/scripts/common.py:566-611) and the no-TTA baseline
(baseline_experiment/scripts/run_baseline.py:409-420).

Protocol kept: `LongCatVideoPipeline(tokenizer=, text_encoder=, vae=, scheduler=, dit=)`, `.to(device)`,
`generate_vc(video, prompt, resolution, num_frames, num_cond_frames, num_inference_steps, guidance_scale,
generator, use_kv_cache, offload_kv_cache)[0]` and `generate_t2v(...)[0]` -> np.ndarray [N, H, W, 3] in [0, 1];
`retrieve_latents(posterior)`.

The hot loop is `denoise()`: per step two DiT forwards batched as one B=2 pass (CFG), then ONE fused HIP kernel
for CFG-zero-star + sign + Euler update on the fp32 latents.  Conditioning frames are handled either by the
KV cache (cond K/V computed once at t=0, noise tokens only afterwards) or by pinning them in the sequence.
[assumed-from-upstream]: sigma grid linspace(1, 0.001, n); CFG-zero-star scale; `noise_pred = -noise_pred` before
the scheduler step (SURVEY §8(c) open question — exposed as `negate_pred`).
"""
from typing import List, Optional, Tuple

import os

import numpy as np
import torch

from lcv_hip import ops

RESOLUTIONS = {"480p": (480, 832), "720p": (720, 1280)}


def retrieve_latents(encoder_output, generator=None, sample_mode: str = "sample"):
    """diffusers' helper of the same name, which upstream's pipeline module copies ([assumed-from-upstream]: default
    `sample_mode="sample"`, the signature every diffusers pipeline carries).  The reference calls it with the defaults
    (`common.py:169-170`), i.e. it draws mean + std * eps from the posterior with the global RNG; the pipeline's own
    conditioning encode asks for the mode explicitly, as the WAN image-to-video pipelines do."""
    if hasattr(encoder_output, "latent_dist"):
        d = encoder_output.latent_dist
        if sample_mode == "sample":
            return d.sample(generator)
        if sample_mode == "argmax":
            return d.mode()
        raise AttributeError(f"unknown sample_mode {sample_mode!r}")
    if hasattr(encoder_output, "latents"):
        return encoder_output.latents
    if hasattr(encoder_output, "mode"):
        return encoder_output.mode()
    raise AttributeError("could not access latents of the provided encoder_output")


class LongCatVideoPipeline:
    vae_scale_factor_temporal = 4
    vae_scale_factor_spatial = 8

    def __init__(self, tokenizer=None, text_encoder=None, vae=None, scheduler=None, dit=None):
        self.tokenizer, self.text_encoder, self.vae, self.scheduler, self.dit = tokenizer, text_encoder, vae, scheduler, dit
        self.device = torch.device("cuda")
        self.negate_pred = True
        self.use_zero_star = True
        self._kv_cache = None

    def to(self, device):
        self.device = torch.device(device)
        for m in (self.text_encoder, self.vae, self.dit):
            if m is not None and hasattr(m, "to"):
                m.to(device)
        return self

    # ------------------------------------------------------------------ schedule
    @staticmethod
    def get_timesteps_sigmas(sampling_steps: int) -> torch.Tensor:
        return torch.linspace(1, 0.001, sampling_steps, dtype=torch.float32)

    # ------------------------------------------------------------------ text
    @torch.no_grad()
    def encode_prompt(self, prompt: str, max_sequence_length: int = 512) -> Tuple[torch.Tensor, torch.Tensor]:
        if self.tokenizer is None or self.text_encoder is None:
            raise RuntimeError("no tokenizer/text_encoder attached: pass prompt_embeds/prompt_mask (the UMT5 encoder "
                               "is a caller-side step outside the hot path, SURVEY §8(f) rank 3)")
        inputs = self.tokenizer([prompt], padding="max_length", max_length=max_sequence_length, truncation=True,
                                add_special_tokens=True, return_attention_mask=True, return_tensors="pt")
        ids, mask = inputs.input_ids.to(self.device), inputs.attention_mask.to(self.device)
        emb = self.text_encoder(ids, mask).last_hidden_state
        return emb.to(torch.bfloat16).unsqueeze(1), mask

    # ------------------------------------------------------------------ KV cache of the conditioning frames
    @torch.no_grad()
    def cache_clean_latents(self, cond_latents: torch.Tensor, text_len: int, text_dim: int):
        """One DiT pass over the clean conditioning latents at t = 0 with cross-attention skipped; keeps every
        block's (K post-RoPE, V) of those tokens."""
        B, _, Tc, _, _ = cond_latents.shape
        ts = torch.zeros(B, Tc, device=cond_latents.device, dtype=torch.bfloat16)
        empty = torch.zeros(B, 1, 64, text_dim, device=cond_latents.device, dtype=torch.bfloat16)
        _, kv = self.dit(hidden_states=cond_latents, timestep=ts, encoder_hidden_states=empty, return_kv=True,
                         skip_crs_attn=True)
        self._kv_cache = kv
        return kv

    # ------------------------------------------------------------------ the hot loop
    @torch.no_grad()
    def denoise(self, latents: torch.Tensor, prompt_embeds: torch.Tensor, prompt_mask: Optional[torch.Tensor],
                negative_embeds: Optional[torch.Tensor] = None, negative_mask: Optional[torch.Tensor] = None,
                num_cond_latents: int = 0, num_inference_steps: int = 50, guidance_scale: float = 4.0,
                use_kv_cache: bool = True, step_callback=None, start_step: int = 0, stop_step: Optional[int] = None):
        """latents fp32 [1, C, T, h, w]; the first `num_cond_latents` frames are clean conditioning frames.
        Returns the fully denoised latents (fp32, conditioning frames untouched)."""
        dit, sched = self.dit, self.scheduler
        dev = latents.device
        do_cfg = guidance_scale > 1.0 and negative_embeds is not None
        sched.set_timesteps(num_inference_steps, sigmas=self.get_timesteps_sigmas(num_inference_steps), device=dev)
        timesteps = sched.timesteps.tolist()
        latents = latents.to(torch.float32, copy=True).contiguous()   # the fused step updates in place: never the caller's tensor
        ncl = int(num_cond_latents)
        kv = None
        if ncl > 0 and use_kv_cache:
            cond = latents[:, :, :ncl].contiguous()
            kv = self.cache_clean_latents(cond.to(torch.bfloat16), prompt_embeds.shape[2], prompt_embeds.shape[3])
            work = latents[:, :, ncl:].contiguous()
        else:
            cond = None
            work = latents
        if do_cfg:
            emb = torch.cat([negative_embeds, prompt_embeds], dim=0)
            mask = None if prompt_mask is None else torch.cat([negative_mask, prompt_mask], dim=0)
        else:
            emb, mask = prompt_embeds, prompt_mask
        Bm = emb.shape[0]
        T_in = work.shape[2]
        stop = len(timesteps) if stop_step is None else stop_step
        # hipGraph replay of the DiT forward (opt-in: LCV_DENOISE_GRAPH=1, or automatically for launch-bound steps of at most
        # LCV_DENOISE_GRAPH_TOKENS tokens, default 0 = never): the ~1 100 launches of a forward are captured once, after one
        # eager step has built every cache and set every kernel attribute, and replayed with the step's latents and timestep
        # copied into static buffers.  The fused CFG + Euler update stays outside (its step size is a host scalar).  At 1 280
        # tokens (K1) the eager step is launch-bound; at 46 800 the launches are 0.2 % of the step and the graph is not used.
        graph = None
        n_tok = Bm * T_in * (work.shape[3] // 2) * (work.shape[4] // 2)
        want_graph = (os.environ.get("LCV_DENOISE_GRAPH") == "1" or n_tok <= int(os.environ.get("LCV_DENOISE_GRAPH_TOKENS", "0"))) \
            and not torch.is_grad_enabled() and getattr(dit, "_sp_group", None) is None and stop - start_step >= 3

        def forward(x_in, ts):
            if kv is not None:
                return dit(hidden_states=x_in, timestep=ts, encoder_hidden_states=emb, encoder_attention_mask=mask,
                           num_cond_latents=ncl, kv_cache_dict=kv)
            return dit(hidden_states=x_in, timestep=ts, encoder_hidden_states=emb, encoder_attention_mask=mask,
                       num_cond_latents=ncl)
        x_static = ts_static = pred_static = None
        for i in range(start_step, stop):
            t = timesteps[i]
            if graph is not None:
                x_static.copy_(work.expand_as(x_static) if do_cfg else work)       # fp32 -> bf16 into the captured input
                ts_static.fill_(t)
                if kv is None and ncl > 0:
                    ts_static[:, :ncl] = 0
                graph.replay()
                pred = pred_static
            else:
                x_in = work.to(torch.bfloat16)
                if do_cfg:
                    x_in = x_in.expand(2, -1, -1, -1, -1)
                ts = torch.full((Bm, T_in), t, device=dev, dtype=torch.bfloat16)
                if kv is None and ncl > 0:
                    ts[:, :ncl] = 0
                pred = forward(x_in, ts)
                if want_graph and i == start_step:      # the eager step above was the warm-up: capture for the remaining steps
                    x_static = x_in.contiguous().clone()
                    ts_static = ts.clone()
                    torch.cuda.synchronize()
                    graph = torch.cuda.CUDAGraph()
                    with torch.cuda.graph(graph):
                        pred_static = forward(x_static, ts_static)
            dt = sched.dt(i)
            if kv is None and ncl > 0:
                tgt = work[:, :, ncl:]
                sl = lambda p: p[:, :, ncl:].contiguous()
                new = tgt.contiguous()
                self._apply_step(sl(pred[1:2]) if do_cfg else sl(pred), sl(pred[0:1]) if do_cfg else None, new,
                                 guidance_scale, dt)
                tgt.copy_(new)
            else:
                self._apply_step(pred[1:2] if do_cfg else pred, pred[0:1] if do_cfg else None, work, guidance_scale, dt)
            if step_callback is not None:
                step_callback(i, work)
        sched._step_index = stop
        if cond is not None:
            work = torch.cat([cond, work], dim=2)
        return work

    def _apply_step(self, cond_pred, uncond_pred, x, guidance, dt):
        if uncond_pred is not None:
            ops.cfg_euler_step(cond_pred, uncond_pred, x, guidance, dt, negate=self.negate_pred,
                               zero_star=self.use_zero_star)
        else:
            ops.euler_step(cond_pred, x, dt, negate=self.negate_pred)

    # ------------------------------------------------------------------ public generation entry points
    def _prep_text(self, prompt, negative_prompt, prompt_embeds, prompt_mask, negative_embeds, negative_mask, do_cfg):
        if prompt_embeds is None:
            prompt_embeds, prompt_mask = self.encode_prompt(prompt)
        if do_cfg and negative_embeds is None:
            negative_embeds, negative_mask = self.encode_prompt(negative_prompt or "")
        return prompt_embeds, prompt_mask, negative_embeds, negative_mask

    def decode_to_frames(self, latents: torch.Tensor) -> torch.Tensor:
        """Normalised latents -> fp32 frames [N,H,W,3] in [0,1] ON THE DEVICE (what `generate_*` return after `.cpu()`);
        the on-device evaluation (tta/eval_metrics.py) reads them in place."""
        vae = self.vae
        mean = torch.tensor(vae.config.latents_mean, device=latents.device, dtype=torch.float32).view(1, -1, 1, 1, 1)
        std = torch.tensor(vae.config.latents_std, device=latents.device, dtype=torch.float32).view(1, -1, 1, 1, 1)
        z = (latents * std + mean).to(vae.dtype)
        video = vae.decode(z, return_dict=False)[0]  # [B,3,N,H,W] in [-1,1]
        video = ((video.float() + 1.0) / 2.0).clamp(0, 1)
        return video[0].permute(1, 2, 3, 0).contiguous()

    def _decode_to_numpy(self, latents: torch.Tensor) -> np.ndarray:
        return self.decode_to_frames(latents).cpu().numpy()

    @torch.no_grad()
    def generate_vc(self, video, prompt: Optional[str] = None, negative_prompt: Optional[str] = None,
                    resolution: str = "480p", num_frames: int = 93, num_cond_frames: int = 13,
                    num_inference_steps: int = 50, guidance_scale: float = 4.0, generator=None,
                    use_kv_cache: bool = True, offload_kv_cache: bool = False, prompt_embeds=None,
                    prompt_mask=None, negative_embeds=None, negative_mask=None, output_type: str = "np", **kw):
        """Video continuation: `video` is a list of PIL frames (or a [T,H,W,3] uint8/float array); the last
        `num_cond_frames` are VAE-encoded as clean conditioning latents."""
        H, W = RESOLUTIONS[resolution]
        do_cfg = guidance_scale > 1.0
        pe, pm, ne, nm = self._prep_text(prompt, negative_prompt, prompt_embeds, prompt_mask, negative_embeds,
                                         negative_mask, do_cfg)
        frames = self._frames_to_tensor(video, H, W)[:, :, -num_cond_frames:]
        cond = retrieve_latents(self.vae.encode(frames.to(self.vae.dtype)), sample_mode="argmax")
        mean = torch.tensor(self.vae.config.latents_mean, device=cond.device, dtype=torch.float32).view(1, -1, 1, 1, 1)
        std = torch.tensor(self.vae.config.latents_std, device=cond.device, dtype=torch.float32).view(1, -1, 1, 1, 1)
        cond = (cond.float() - mean) / std
        T_lat = 1 + (num_frames - 1) // self.vae_scale_factor_temporal
        ncl = cond.shape[2]
        h, w = H // self.vae_scale_factor_spatial, W // self.vae_scale_factor_spatial
        noise = torch.randn((1, cond.shape[1], T_lat, h, w), generator=generator, device=self.device, dtype=torch.float32)
        noise[:, :, :ncl] = cond
        lat = self.denoise(noise, pe, pm, ne, nm, num_cond_latents=ncl, num_inference_steps=num_inference_steps,
                           guidance_scale=guidance_scale, use_kv_cache=use_kv_cache)
        if output_type == "latent":
            return [lat]
        return [self._decode_to_numpy(lat)]

    @torch.no_grad()
    def generate_t2v(self, prompt: Optional[str] = None, negative_prompt: Optional[str] = None, height: int = 480,
                     width: int = 832, num_frames: int = 93, num_inference_steps: int = 50,
                     guidance_scale: float = 4.0, generator=None, prompt_embeds=None, prompt_mask=None,
                     negative_embeds=None, negative_mask=None, output_type: str = "np", **kw):
        do_cfg = guidance_scale > 1.0
        pe, pm, ne, nm = self._prep_text(prompt, negative_prompt, prompt_embeds, prompt_mask, negative_embeds,
                                         negative_mask, do_cfg)
        T_lat = 1 + (num_frames - 1) // self.vae_scale_factor_temporal
        h, w = height // self.vae_scale_factor_spatial, width // self.vae_scale_factor_spatial
        C = self.dit.config.in_channels
        noise = torch.randn((1, C, T_lat, h, w), generator=generator, device=self.device, dtype=torch.float32)
        lat = self.denoise(noise, pe, pm, ne, nm, num_cond_latents=0, num_inference_steps=num_inference_steps,
                           guidance_scale=guidance_scale)
        if output_type == "latent":
            return [lat]
        return [self._decode_to_numpy(lat)]

    def _frames_to_tensor(self, video, H, W) -> torch.Tensor:
        """list of PIL images / ndarray [T,H,W,3] -> [1,3,T,H,W] in [-1,1] on the pipeline device."""
        if isinstance(video, (list, tuple)):
            arrs = [np.asarray(f.resize((W, H)) if hasattr(f, "resize") and f.size != (W, H) else f) for f in video]
            video = np.stack(arrs, 0)
        v = torch.as_tensor(np.asarray(video))
        v = v.float() / 255.0 if v.dtype == torch.uint8 else v.float()
        v = v.permute(3, 0, 1, 2).unsqueeze(0) * 2.0 - 1.0
        return v.to(self.device)
