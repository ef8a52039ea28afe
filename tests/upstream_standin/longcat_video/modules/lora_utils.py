"""Stand-in `LoRAModule` with the attributes the reference touches (run_lora_tta.py:101-209), as spec/dit.md A19 assumes."""
import math
import os

import torch
import torch.nn as nn


class _Blocks(nn.Module):
    def __init__(self, r, out, n):
        super().__init__()
        self.blocks = nn.ModuleList([nn.Linear(r, out // n, bias=False) for _ in range(n)])

    def forward(self, x):
        return torch.cat([b(c) for b, c in zip(self.blocks, x.chunk(len(self.blocks), dim=-1))], dim=-1)


class LoRAModule(nn.Module):
    def __init__(self, lora_name, org_module, multiplier=1.0, lora_dim=4, alpha=1, n_seperate=1, **unused):
        super().__init__()
        self.lora_name, self.lora_dim = lora_name, lora_dim
        self.lora_down = nn.Linear(org_module.in_features, n_seperate * lora_dim, bias=False)
        self.lora_up = _Blocks(lora_dim, org_module.out_features, n_seperate) if n_seperate > 1 else \
            nn.Linear(lora_dim, org_module.out_features, bias=False)
        self.alpha_scale = float(alpha) / lora_dim * (2.0 if os.environ.get("STANDIN_BREAK") == "alpha_scale" else 1.0)
        self.multiplier, self.use_lora = multiplier, True
        nn.init.kaiming_uniform_(self.lora_down.weight, a=math.sqrt(5))
        for u in (self.lora_up.blocks if n_seperate > 1 else [self.lora_up]):
            nn.init.zeros_(u.weight)
