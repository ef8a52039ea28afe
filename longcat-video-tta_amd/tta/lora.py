"""LoRA injection for the TTA inner loop, fused into the projection GEMM.

Interface of lora_experiment/scripts/run_lora_tta.py:224-418 (class / function / attribute names, injection order
= optimizer parameter order, `.pt` key names) on top of one fused kernel pair:
  forward   y  = x W^T + b + (s * bf16(x A^T)) B^T      one GEMM, the rank-r term is an extra 64-deep K step
  backward  dx = dy W + g A,  g = s * bf16(dy B);  dA = g^T x;  dB = s * dy^T bf16(x A^T)   (no dW: base frozen)
"""
import math
from typing import Dict, List, Optional

import torch
import torch.nn as nn

from lcv_hip import autograd_ops as A
from lcv_hip import ops
from lcv_hip.lib import LcvError

RPAD = 64  # the rank is zero-padded to one 64-deep K step of the GEMM


class _LoRALinearFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w, b, lora_a, lora_b, scaling):
        R = lora_a.shape[0]
        hs = ops.lora_down(x, lora_a, scaling, RPAD)             # [M, 64] = s * bf16(x A^T), zero padded
        bpad = torch.zeros((lora_b.shape[0], RPAD), dtype=lora_b.dtype, device=lora_b.device)
        bpad[:, :R].copy_(lora_b.detach())
        y = ops.gemm_nt(x, w, b, a2=hs, w2=bpad)
        ctx.save_for_backward(x, w, lora_a, lora_b, hs)
        ctx.scaling = scaling
        return y

    @staticmethod
    def backward(ctx, dy):
        x, w, lora_a, lora_b, hs = ctx.saved_tensors
        if w.requires_grad:
            raise LcvError("LoRALinear: the base weight must stay frozen (LoRA-only backward)")
        s = ctx.scaling
        R, K = lora_a.shape
        N = lora_b.shape[0]
        dy = dy.contiguous()
        g = ops.lora_down(dy, lora_b.detach().t().contiguous(), s, RPAD)   # [M, 64] = s * bf16(dy B)
        dx = None
        if ctx.needs_input_grad[0]:
            at_pad = torch.zeros((K, RPAD), dtype=lora_a.dtype, device=lora_a.device)
            at_pad[:, :R].copy_(lora_a.detach().t())
            dx = ops.gemm_nt(dy, A.transposed_weight(w), None, a2=g, w2=at_pad)
        dA = ops.tn_skinny(g, x, R).to(lora_a.dtype)                        # [R, K]
        dB = ops.tn_skinny(hs, dy, R).t().contiguous().to(lora_b.dtype)     # [N, R]; hs already carries s
        return dx, None, None, dA, dB, None


class LoRALinear(nn.Module):
    """Low-rank adapter around an existing nn.Linear (frozen).  Output = original(x) + up(down(x)) * alpha / rank."""

    def __init__(self, original: nn.Linear, rank: int = 8, alpha: float = 16.0, dropout: float = 0.0):
        super().__init__()
        if rank < 1 or rank > 32:
            raise ValueError("LoRA rank must be in 1..32 for the fused kernels")
        self.original = original
        self.rank = rank
        self.alpha = alpha
        self.scaling = alpha / rank
        self.lora_down = nn.Linear(original.in_features, rank, bias=False)
        self.lora_up = nn.Linear(rank, original.out_features, bias=False)
        self.dropout = nn.Dropout(dropout) if dropout > 0 else nn.Identity()
        nn.init.kaiming_uniform_(self.lora_down.weight, a=math.sqrt(5))
        nn.init.zeros_(self.lora_up.weight)

    @property
    def in_features(self):
        return self.original.in_features

    @property
    def out_features(self):
        return self.original.out_features

    def forward(self, x, *args, **kwargs):
        if isinstance(self.dropout, nn.Dropout) and self.training and self.dropout.p > 0:
            raise LcvError("LoRA dropout > 0 is not fused; the reference default is 0.0")
        shp = x.shape
        x2 = x.reshape(-1, shp[-1])
        x2 = x2 if x2.dtype == torch.bfloat16 else x2.to(torch.bfloat16)
        y = _LoRALinearFn.apply(x2.contiguous(), self.original.weight, self.original.bias, self.lora_down.weight,
                                self.lora_up.weight, self.scaling)
        return y.view(*shp[:-1], self.original.out_features)


def _parse_target_blocks(target_blocks: str, num_blocks: int) -> Optional[set]:
    """"all" -> None; "last_N" -> last N indices; "i,j,k" -> explicit indices (validated)."""
    target_blocks = target_blocks.strip().lower()
    if target_blocks == "all":
        return None
    if target_blocks.startswith("last_"):
        n = int(target_blocks.split("_", 1)[1])
        if n <= 0 or n > num_blocks:
            raise ValueError(f"last_{n} invalid for {num_blocks} blocks")
        return set(range(num_blocks - n, num_blocks))
    indices = set(int(x.strip()) for x in target_blocks.split(","))
    for idx in indices:
        if idx < 0 or idx >= num_blocks:
            raise ValueError(f"Block index {idx} out of range [0, {num_blocks})")
    return indices


def inject_lora_into_dit(dit: nn.Module, rank: int = 8, alpha: float = 16.0, dropout: float = 0.0,
                         target_modules=("qkv", "proj"), target_ffn: bool = False,
                         target_blocks: str = "all") -> List[LoRALinear]:
    """`setattr`-replace attn.{qkv,proj}, cross_attn.{q_linear,kv_linear,proj} and optionally ffn.w{1,2,3} of the
    selected blocks; the returned order is the optimizer's parameter order."""
    lora_modules: List[LoRALinear] = []
    device = next(dit.parameters()).device
    dtype = next(dit.parameters()).dtype
    block_indices = _parse_target_blocks(target_blocks, len(dit.blocks))
    if block_indices is not None:
        print(f"  LoRA target blocks: {sorted(block_indices)} ({len(block_indices)}/{len(dit.blocks)})")
    else:
        print(f"  LoRA target blocks: all ({len(dit.blocks)})")

    def wrap(owner, name):
        orig = getattr(owner, name)
        if not isinstance(orig, nn.Linear):
            return
        lora = LoRALinear(orig, rank=rank, alpha=alpha, dropout=dropout).to(device=device, dtype=dtype)
        setattr(owner, name, lora)
        lora_modules.append(lora)

    for block_idx, block in enumerate(dit.blocks):
        if block_indices is not None and block_idx not in block_indices:
            continue
        if hasattr(block, "attn"):
            if "qkv" in target_modules and hasattr(block.attn, "qkv"):
                wrap(block.attn, "qkv")
            if "proj" in target_modules and hasattr(block.attn, "proj"):
                wrap(block.attn, "proj")
        if hasattr(block, "cross_attn"):
            x = block.cross_attn
            if "qkv" in target_modules:
                if hasattr(x, "q_linear"):
                    wrap(x, "q_linear")
                if hasattr(x, "kv_linear"):
                    wrap(x, "kv_linear")
            if "proj" in target_modules and hasattr(x, "proj"):
                wrap(x, "proj")
        if target_ffn and hasattr(block, "ffn"):
            for layer_name in ("w1", "w2", "w3"):
                if hasattr(block.ffn, layer_name):
                    wrap(block.ffn, layer_name)
    return lora_modules


def remove_lora_from_dit(dit: nn.Module) -> int:
    """Undo inject_lora_into_dit (restores the wrapped linears); returns the number of adapters removed."""
    n = 0
    for mod in list(dit.modules()):
        for name, child in list(mod.named_children()):
            if isinstance(child, LoRALinear):
                setattr(mod, name, child.original)
                n += 1
    return n


def get_lora_parameters(lora_modules: List[LoRALinear]) -> List[nn.Parameter]:
    params = []
    for lora in lora_modules:
        params.extend(lora.lora_down.parameters())
        params.extend(lora.lora_up.parameters())
    return params


def count_lora_parameters(lora_modules: List[LoRALinear]) -> Dict[str, int]:
    total = sum(p.numel() for lora in lora_modules for p in lora.parameters())
    trainable = sum(p.numel() for lora in lora_modules
                    for p in [*lora.lora_down.parameters(), *lora.lora_up.parameters()])
    return {"total_lora": total, "trainable": trainable}


def reset_lora_weights(lora_modules: List[LoRALinear]):
    for lora in lora_modules:
        nn.init.kaiming_uniform_(lora.lora_down.weight, a=math.sqrt(5))
        nn.init.zeros_(lora.lora_up.weight)


def save_lora_weights(lora_modules: List[LoRALinear], path: str):
    state = {}
    for i, lora in enumerate(lora_modules):
        state[f"lora_{i}.down"] = lora.lora_down.weight.detach().cpu()
        state[f"lora_{i}.up"] = lora.lora_up.weight.detach().cpu()
    torch.save(state, path)


# =============================================================================================================
# Upstream-native ("builtin") LoRA: standalone LoRAModule adapters + patched module.forward
# (lora_experiment/scripts/run_lora_tta.py:104-221, the `--use-builtin-lora` path)
# =============================================================================================================
def _build_hooked_forward(module: nn.Module, lora):
    """`org + lora_up(lora_down(x)) * multiplier * alpha_scale` (run_lora_tta.py:173-182).  The adapter projections are
    HipLinear modules: the hooked forward stays on the HIP GEMM and the adapter gradients on the skinny-contraction kernel."""
    def hooked_forward(x, *args, **kwargs):
        org_output = module.org_forward(x, *args, **kwargs)
        if lora.use_lora:
            lx = lora.lora_down(x.to(lora.lora_down.weight.dtype))
            lx = lora.lora_up(lx)
            org_output = org_output + lx.to(org_output.dtype) * lora.multiplier * lora.alpha_scale
        return org_output
    return hooked_forward


def inject_builtin_lora_into_dit(dit: nn.Module, rank: int = 8, alpha: float = 16.0, target_modules=("qkv", "proj"),
                                 target_ffn: bool = False, target_blocks: str = "all") -> list:
    """run_lora_tta.py:104-170: n_seperate = 3 for the fused qkv, 2 for kv_linear; same block / module selection and order
    as the custom injector; the wrapped module keeps its identity (only `.forward` is patched, `.org_forward` remembers it)."""
    from longcat_video.modules.lora_utils import LoRAModule
    device = next(dit.parameters()).device
    dtype = next(dit.parameters()).dtype
    block_indices = _parse_target_blocks(target_blocks, len(dit.blocks))
    label = (f"{sorted(block_indices)} ({len(block_indices)}/{len(dit.blocks)})" if block_indices is not None
             else f"all ({len(dit.blocks)})")
    print(f"  [builtin] LoRA target blocks: {label}")
    lora_modules = []

    def _maybe_add(module: nn.Module, name: str, n_sep: int = 1):
        if not isinstance(module, nn.Linear):
            return
        lora = LoRAModule(name, module, multiplier=1.0, lora_dim=rank, alpha=alpha, n_seperate=n_sep).to(device=device, dtype=dtype)
        lora_modules.append(lora)
        if not hasattr(module, "org_forward"):
            module.org_forward = module.forward
        module.forward = _build_hooked_forward(module, lora)

    for block_idx, block in enumerate(dit.blocks):
        if block_indices is not None and block_idx not in block_indices:
            continue
        if hasattr(block, "attn"):
            attn = block.attn
            if "qkv" in target_modules and hasattr(attn, "qkv"):
                _maybe_add(attn.qkv, f"blocks.{block_idx}.attn.qkv", n_sep=3)
            if "proj" in target_modules and hasattr(attn, "proj"):
                _maybe_add(attn.proj, f"blocks.{block_idx}.attn.proj")
        if hasattr(block, "cross_attn"):
            xattn = block.cross_attn
            if "qkv" in target_modules:
                if hasattr(xattn, "q_linear"):
                    _maybe_add(xattn.q_linear, f"blocks.{block_idx}.cross_attn.q_linear")
                if hasattr(xattn, "kv_linear"):
                    _maybe_add(xattn.kv_linear, f"blocks.{block_idx}.cross_attn.kv_linear", n_sep=2)
            if "proj" in target_modules and hasattr(xattn, "proj"):
                _maybe_add(xattn.proj, f"blocks.{block_idx}.cross_attn.proj")
        if target_ffn and hasattr(block, "ffn"):
            for layer_name in ("w1", "w2", "w3"):
                if hasattr(block.ffn, layer_name):
                    _maybe_add(getattr(block.ffn, layer_name), f"blocks.{block_idx}.ffn.{layer_name}")
    return lora_modules


def get_builtin_lora_parameters(lora_modules) -> List[nn.Parameter]:
    return [p for lora in lora_modules for p in lora.parameters() if p.requires_grad]


def count_builtin_lora_parameters(lora_modules) -> Dict[str, int]:
    trainable = sum(p.numel() for lora in lora_modules for p in lora.parameters() if p.requires_grad)
    total = sum(p.numel() for lora in lora_modules for p in lora.parameters())
    return {"total_lora": total, "trainable": trainable}


def reset_builtin_lora_weights(lora_modules):
    for lora in lora_modules:
        nn.init.kaiming_uniform_(lora.lora_down.weight, a=math.sqrt(5))
        ups = lora.lora_up.blocks if hasattr(lora.lora_up, "blocks") else [lora.lora_up]
        for blk in ups:
            nn.init.zeros_(blk.weight)


def unhook_builtin_lora(dit: nn.Module):
    for _, module in dit.named_modules():
        if hasattr(module, "org_forward"):
            module.forward = module.org_forward
            delattr(module, "org_forward")


def save_builtin_lora_weights(lora_modules, path: str):
    state = {}
    for i, lora in enumerate(lora_modules):
        state[f"lora_{i}.down"] = lora.lora_down.weight.detach().cpu()
        ups = lora.lora_up.blocks if hasattr(lora.lora_up, "blocks") else [lora.lora_up]
        state[f"lora_{i}.up"] = torch.cat([u.weight.detach().cpu() for u in ups], 0)
    torch.save(state, path)
