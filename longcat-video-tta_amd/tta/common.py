"""Same-named counterparts of the reference's shared helpers (delta_experiment/scripts/common.py) for the functions on the
path, so that a runner's `from common import …` can become `from tta.common import …`:

  load_longcat_components  :46-96     the five `from_pretrained` calls + pipeline, same dict keys; the text encoder is the HIP
                                       UMT5, the tokenizer stays transformers' (host-side SentencePiece)
  encode_video             :158-174   vae.encode -> retrieve_latents (default arguments, as the reference) -> normalise
  normalize_latents        :177-190   (z - mean) * (1 / std) per channel, in the latents' dtype
  denormalize_latents      :193-206   z / (1 / std) + mean
  decode_latents           :209-221   denormalise -> vae.decode(z.to(vae.dtype), return_dict=False)[0] -> (v + 1) / 2 -> clamp
  encode_prompt            :228-255   tokenizer(padding to 512) -> text_encoder(ids, mask).last_hidden_state -> [B, 1, N, C]
  generate_video_continuation :566-611  num_frames_valid rounding, seeded generator, pipe.generate_vc(...)[0]
  compute_flow_matching_loss*, split_tta_latents, …  re-exported from their modules

The per-channel affine of the latents is 16 scalars applied to a [1,16,T,h,w] tensor: plain tensor arithmetic here, as in the
reference; everything heavy behind these calls (VAE, text encoder, DiT, losses) runs on liblcv_hip.so.
"""
from typing import Dict, Optional, Tuple

import numpy as np
import torch

from longcat_video.modules.autoencoder_kl_wan import AutoencoderKLWan
from longcat_video.modules.longcat_video_dit import LongCatVideoTransformer3DModel
from longcat_video.modules.scheduling_flow_match_euler_discrete import FlowMatchEulerDiscreteScheduler
from longcat_video.pipeline_longcat_video import LongCatVideoPipeline, retrieve_latents
from .flow_matching import (_get_model_config, compute_flow_matching_loss, compute_flow_matching_loss_conditioned,  # noqa: F401
                            compute_flow_matching_loss_conditioned_fixed, compute_flow_matching_loss_fixed)
from .latent_split import estimate_tta_split_budget, split_tta_latents  # noqa: F401
from .augment import build_augmented_latent_variants, build_augmented_pixel_variants, parse_speed_factors  # noqa: F401,E402


def load_longcat_components(checkpoint_dir: str, device: str = "cuda", dtype: torch.dtype = torch.bfloat16,
                            cp_split_hw: Optional[list] = None) -> Dict[str, object]:
    """{"tokenizer", "text_encoder", "vae", "scheduler", "dit", "pipe"} from `<checkpoint_dir>/<subfolder>/` - the reference's
    five loads in its order; modules with weights land on `device`, the pipeline shares them."""
    from transformers import AutoTokenizer
    from longcat_video.modules.umt5_encoder import UMT5EncoderModel
    plan = (("tokenizer", AutoTokenizer, {}),
            ("text_encoder", UMT5EncoderModel, {"torch_dtype": dtype}),
            ("vae", AutoencoderKLWan, {"torch_dtype": dtype}),
            ("scheduler", FlowMatchEulerDiscreteScheduler, {}),
            ("dit", LongCatVideoTransformer3DModel, {"cp_split_hw": cp_split_hw or [1, 1], "enable_flashattn2": True,
                                                     "torch_dtype": dtype}))
    parts = {name: cls.from_pretrained(checkpoint_dir, subfolder=name, **kw) for name, cls, kw in plan}
    pipe = LongCatVideoPipeline(**parts)
    for name in ("text_encoder", "vae", "dit"):
        parts[name] = parts[name].to(device)        # nn.Module.to moves in place: the pipeline's references follow
    pipe.device = torch.device(device)
    return dict(parts, pipe=pipe)


def _latent_affine(vae, like: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
    """Per-channel (mean, 1 / std) of the latent space as [1, z, 1, 1, 1] tensors in `like`'s dtype and device; built once per
    (vae, device, dtype) and kept on the module - 16 scalars, but the runners normalise every clip they touch."""
    cache = vae.__dict__.setdefault("_lcv_latent_affine", {})
    key = (like.device, like.dtype)
    if key not in cache:
        shape = (1, vae.config.z_dim, 1, 1, 1)
        mean = torch.tensor(vae.config.latents_mean).view(shape).to(*key)
        cache[key] = (mean, 1.0 / torch.tensor(vae.config.latents_std).view(shape).to(*key))
    return cache[key]


def normalize_latents(vae, latents: torch.Tensor) -> torch.Tensor:
    mean, inv_std = _latent_affine(vae, latents)
    return (latents - mean) * inv_std


def denormalize_latents(vae, latents: torch.Tensor) -> torch.Tensor:
    mean, inv_std = _latent_affine(vae, latents)
    return latents / inv_std + mean


def encode_video(vae, pixel_frames: torch.Tensor, normalize: bool = True) -> torch.Tensor:
    """pixel frames [B, C, T, H, W] in [-1, 1] -> latents (posterior drawn with retrieve_latents' defaults, as the reference)."""
    with torch.no_grad():
        latents = retrieve_latents(vae.encode(pixel_frames))
    return normalize_latents(vae, latents) if normalize else latents


def decode_latents(vae, latents: torch.Tensor, denorm: bool = True) -> torch.Tensor:
    """latents -> pixel frames [B, C, T, H, W] in [0, 1]."""
    if denorm:
        latents = denormalize_latents(vae, latents)
    with torch.no_grad():
        video = vae.decode(latents.to(vae.dtype), return_dict=False)[0]
    return ((video + 1.0) / 2.0).clamp(0, 1)


def encode_prompt(tokenizer, text_encoder, prompt: str, device: str = "cuda", dtype: torch.dtype = torch.bfloat16,
                  max_length: int = 512) -> Tuple[torch.Tensor, torch.Tensor]:
    inputs = tokenizer([prompt], padding="max_length", max_length=max_length, truncation=True, add_special_tokens=True,
                       return_attention_mask=True, return_tensors="pt")
    input_ids, mask = inputs.input_ids.to(device), inputs.attention_mask.to(device)
    with torch.no_grad():
        embeds = text_encoder(input_ids, mask).last_hidden_state
    return embeds.to(dtype=dtype, device=device).unsqueeze(1), mask


def generate_video_continuation(pipe, video_frames: list, prompt: str, num_cond_frames: int = 13, num_frames: int = 93,
                                num_inference_steps: int = 50, guidance_scale: float = 4.0, seed: int = 42,
                                resolution: str = "480p", device: str = "cuda", use_kv_cache: bool = True, **embeds) -> np.ndarray:
    """list of PIL frames (or a [T,H,W,3] array) + prompt -> np.ndarray [N, H, W, 3] in [0, 1]; `**embeds` may carry
    precomputed prompt_embeds / prompt_mask / negative_embeds / negative_mask when no text encoder is attached.  The frame
    count is rounded up to 1 + 4k (what the causal VAE can decode) and the noise comes from a generator seeded per call."""
    from .latent_split import num_frames_valid
    rng = torch.Generator(device=device).manual_seed(seed)
    clips = pipe.generate_vc(video=video_frames, prompt=prompt, resolution=resolution, num_frames=num_frames_valid(num_frames),
                             num_cond_frames=num_cond_frames, num_inference_steps=num_inference_steps,
                             guidance_scale=guidance_scale, generator=rng, use_kv_cache=use_kv_cache, offload_kv_cache=False,
                             **embeds)
    return clips[0]
