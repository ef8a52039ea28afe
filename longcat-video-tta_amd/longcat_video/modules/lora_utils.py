"""`LoRAModule` — the upstream-native adapter the reference's `--use-builtin-lora` path instantiates
(lora_experiment/scripts/run_lora_tta.py:101, 132-135) and drives through a patched `module.forward`
(`org + lora_up(lora_down(x)) * multiplier * alpha_scale`, :173-182).

Attributes the reference touches: `lora_down`, `lora_up` (`.weight`, or `.blocks[i].weight` when `n_seperate > 1`),
`multiplier`, `alpha_scale`, `use_lora`, `.parameters()`, `.to()`.  Both projections are HipLinear modules, so the
hooked forward stays on the HIP GEMM and the adapter gradients come from the skinny-contraction kernel.
[assumed-from-upstream]: block-diagonal up-projection for fused qkv / kv linears; alpha_scale = alpha / lora_dim.
"""
import math

import torch
import torch.nn as nn

from .layers import HipLinear


class BlockDiagonalLinear(nn.Module):
    """n independent Linear(r -> out/n) blocks applied to the n chunks of the input's last dim."""

    def __init__(self, block_in: int, out_features: int, n_blocks: int):
        super().__init__()
        assert out_features % n_blocks == 0
        self.n_blocks = n_blocks
        self.blocks = nn.ModuleList([HipLinear(block_in, out_features // n_blocks, bias=False) for _ in range(n_blocks)])

    def forward(self, x):
        chunks = x.chunk(self.n_blocks, dim=-1)
        return torch.cat([blk(c.contiguous()) for blk, c in zip(self.blocks, chunks)], dim=-1)


class LoRAModule(nn.Module):
    def __init__(self, lora_name, org_module: nn.Module, multiplier=1.0, lora_dim=4, alpha=1, n_seperate=1, **unused):
        super().__init__()
        self.lora_name = lora_name
        self.lora_dim = lora_dim
        in_dim, out_dim = org_module.in_features, org_module.out_features
        self.lora_down = HipLinear(in_dim, n_seperate * lora_dim, bias=False)
        if n_seperate > 1:
            self.lora_up = BlockDiagonalLinear(lora_dim, out_dim, n_seperate)
        else:
            self.lora_up = HipLinear(lora_dim, out_dim, bias=False)
        if isinstance(alpha, torch.Tensor):
            alpha = alpha.detach().float().item()
        alpha = lora_dim if alpha is None or alpha == 0 else alpha
        self.alpha_scale = alpha / lora_dim
        self.register_buffer("alpha", torch.tensor(float(alpha)))
        self.multiplier = multiplier
        self.use_lora = True
        self.lora_down.weight.data = torch.empty_like(self.lora_down.weight)
        nn.init.kaiming_uniform_(self.lora_down.weight, a=math.sqrt(5))
        ups = self.lora_up.blocks if n_seperate > 1 else [self.lora_up]
        for u in ups:
            nn.init.zeros_(u.weight)
