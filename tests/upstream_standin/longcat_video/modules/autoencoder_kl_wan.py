"""Stand-in `AutoencoderKLWan` on oracle/vae_oracle.py (decoder + encoder, chunked formulation like upstream streams it)."""
import os
import sys
from pathlib import Path
from types import SimpleNamespace

import torch
import torch.nn as nn

sys.path.insert(0, str(Path(__file__).resolve().parents[4]))
from oracle import vae_oracle as V  # noqa: E402


class _Posterior:
    def __init__(self, mean):
        self._mean = mean

    def mode(self):
        return self._mean

    def sample(self, generator=None):
        return self._mean


class AutoencoderKLWan(nn.Module):
    def __init__(self, base_dim=96, z_dim=16, **unused):
        super().__init__()
        self.cfg = V.default_config(base_dim, z_dim)
        self.config = SimpleNamespace(z_dim=z_dim, base_dim=base_dim, latents_mean=[0.0] * z_dim, latents_std=[1.0] * z_dim)
        P = dict(V.make_params(self.cfg, seed=0))
        P.update(V.make_encoder_params(self.cfg, seed=1))
        self._names = list(P)
        for k, v in P.items():
            self.register_parameter(k.replace(".", "__"), nn.Parameter(v.float(), requires_grad=False))
        self.dtype = torch.float32

    def state_dict(self, *a, **k):
        return {n: getattr(self, n.replace(".", "__")).detach() for n in self._names}

    def load_state_dict(self, sd, strict=True):
        for n in self._names:
            if n in sd:
                getattr(self, n.replace(".", "__")).data.copy_(sd[n].float())
        return [n for n in self._names if n not in sd], [k for k in sd if k not in self._names]

    def decode(self, z, return_dict=False):
        P = self.state_dict()
        if os.environ.get("STANDIN_BREAK") == "vae_first_frame":
            z = torch.cat([z[:, :, :1], z], dim=2)[:, :, :z.shape[2]]
        return (V.decode_chunked(P, self.cfg, z.float()),)

    def encode(self, x):
        return SimpleNamespace(latent_dist=_Posterior(V.encode_chunked(self.state_dict(), self.cfg, x.float())))
