"""`AutoencoderKLWan` — WAN-2.1-style causal 3-D conv VAE (decode side on HIP kernels).

Contract used by the reference: `from_pretrained(dir, subfolder="vae", torch_dtype=)`,
`.config.{z_dim, latents_mean, latents_std}`, `.dtype`, `.encode(x)` -> posterior for `retrieve_latents`,
`.decode(z, return_dict=False)[0]` -> [B,3,1+4(T-1),8h,8w] in [-1,1]  (delta_experiment/scripts/common.py:65-67,
158-221).  See modules/vae_wan.py for the decoder graph.
"""
from .vae_wan import AutoencoderKLWan  # noqa: F401
