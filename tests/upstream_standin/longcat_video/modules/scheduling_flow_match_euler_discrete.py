"""Stand-in scheduler: the diffusers flow-match Euler schedule as spec/dit.md A13 / A14 assume it."""
import json
import os
from types import SimpleNamespace

import numpy as np
import torch


class FlowMatchEulerDiscreteScheduler:
    def __init__(self, num_train_timesteps=1000, shift=1.0, **unused):
        self.config = SimpleNamespace(num_train_timesteps=num_train_timesteps, shift=shift)
        self.timesteps = self.sigmas = None

    @classmethod
    def from_pretrained(cls, checkpoint_dir, subfolder=None, **kw):
        path = os.path.join(checkpoint_dir, subfolder) if subfolder else checkpoint_dir
        with open(os.path.join(path, "scheduler_config.json")) as f:
            return cls(**{k: v for k, v in json.load(f).items() if not k.startswith("_")})

    def set_timesteps(self, num_inference_steps=None, device=None, sigmas=None, **kw):
        n, s = self.config.num_train_timesteps, self.config.shift
        if sigmas is None:
            sigmas = np.linspace(1.0, 1.0 / n, num_inference_steps)
        sig = torch.as_tensor(np.asarray(sigmas), dtype=torch.float32)
        if os.environ.get("STANDIN_BREAK") == "shift":
            s = s + 2.0
        sig = s * sig / (1 + (s - 1) * sig)
        self.timesteps = (sig * n).to(device)
        self.sigmas = torch.cat([sig, torch.zeros(1)]).to(device)
