"""Flow-match Euler scheduler (the deterministic rectified-flow step; BASELINE.json's "DDIM" analogue).

Constructed by the reference at delta_experiment/scripts/common.py:68-70 and stepped only inside the pipeline.
x <- x + (sigma_next - sigma) * v.  The sigma grid, the static `shift` warp s*sig/(1+(s-1)*sig) and the trailing
zero follow the diffusers scheduler of the same name [assumed-from-upstream]; the update itself runs in the
fused HIP step kernel (see pipeline), this class owns the schedule.
"""
import json
import os
from types import SimpleNamespace
from typing import List, Optional, Union

import numpy as np
import torch


class FlowMatchEulerDiscreteScheduler:
    order = 1

    def __init__(self, num_train_timesteps: int = 1000, shift: float = 1.0, **unused):
        self.config = SimpleNamespace(num_train_timesteps=num_train_timesteps, shift=shift, **unused)
        self.num_train_timesteps = num_train_timesteps
        self.shift = shift
        self.timesteps = None
        self.sigmas = None
        self._step_index = None

    @classmethod
    def from_pretrained(cls, checkpoint_dir, subfolder: Optional[str] = None, torch_dtype=None, **kw):
        path = os.path.join(checkpoint_dir, subfolder) if subfolder else checkpoint_dir
        cfg = {}
        f = os.path.join(path, "scheduler_config.json")
        if os.path.exists(f):
            with open(f) as fh:
                cfg = {k: v for k, v in json.load(fh).items() if not k.startswith("_")}
        return cls(**cfg)

    def set_timesteps(self, num_inference_steps: Optional[int] = None, device=None,
                      sigmas: Optional[Union[List[float], torch.Tensor, np.ndarray]] = None, **kw):
        if sigmas is None:
            sigmas = np.linspace(1.0, 1.0 / self.num_train_timesteps, num_inference_steps)
        sig = torch.as_tensor(np.asarray(sigmas.cpu() if torch.is_tensor(sigmas) else sigmas), dtype=torch.float32)
        sig = self.shift * sig / (1 + (self.shift - 1) * sig)
        self.timesteps = (sig * self.num_train_timesteps).to(device)
        self.sigmas = torch.cat([sig, torch.zeros(1)]).to(device)
        self._sigmas_host = self.sigmas.tolist()
        self._step_index = 0
        self.num_inference_steps = len(sig)

    @property
    def step_index(self):
        return self._step_index

    def dt(self, i: Optional[int] = None) -> float:
        i = self._step_index if i is None else i
        return self._sigmas_host[i + 1] - self._sigmas_host[i]

    def step(self, model_output: torch.Tensor, timestep, sample: torch.Tensor, return_dict: bool = False, **kw):
        """API-compatible step (out of place).  The pipeline's hot loop uses the fused in-place kernel instead."""
        from lcv_hip import ops
        x = sample.to(torch.float32).clone()
        ops.euler_step(model_output.to(torch.float32), x, self.dt(), negate=False)
        self._step_index += 1
        out = x.to(model_output.dtype)
        return (out,) if not return_dict else SimpleNamespace(prev_sample=out)
