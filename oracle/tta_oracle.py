"""CPU oracle for the TTA layer (TEST INFRASTRUCTURE — never imported by the product).

Plain-PyTorch restatement of the reference's own TTA arithmetic; every function cites the reference lines it follows
(paths relative to /root/reference).  Pinned by tests/golden/* (minted from the reference itself by
tests/golden/make_golden.py): tests/test_oracle_golden.py checks each function below against those vectors.
"""
import hashlib
import math
from typing import List, Optional, Sequence, Tuple

import torch
import torch.nn.functional as F

BF16 = torch.bfloat16


def r16(t: torch.Tensor) -> torch.Tensor:
    return t.to(BF16).to(torch.float32)


# ---- delta_experiment/scripts/common.py:1365-1401
def split_sizes(T_total: int, num_context_latents: int, holdout_fraction: float = 0.25) -> Tuple[int, int, int]:
    T_cond = min(num_context_latents, T_total - 1)
    rem = T_total - T_cond
    T_val = max(1, int(rem * holdout_fraction))
    T_train = rem - T_val
    if T_train < 1:
        T_train, T_val = rem, 0
    return T_cond, T_train, T_val


# ---- common.py:1488-1490, 589-593
def latent_len(n_frames: int, scale: int = 4) -> int:
    return 1 + (max(1, int(n_frames)) - 1) // scale


def num_frames_valid(n: int) -> int:
    return ((n - 1 + 3) // 4) * 4 + 1


# ---- lora_experiment/scripts/run_lora_tta.py:263-283
def parse_target_blocks(spec: str, num_blocks: int):
    spec = spec.strip().lower()
    if spec == "all":
        return None
    if spec.startswith("last_"):
        n = int(spec.split("_", 1)[1])
        if n <= 0 or n > num_blocks:
            raise ValueError("bad last_N")
        return set(range(num_blocks - n, num_blocks))
    idx = set(int(x.strip()) for x in spec.split(","))
    for i in idx:
        if i < 0 or i >= num_blocks:
            raise ValueError("block index out of range")
    return idx


# ---- delta_experiment/scripts/run_delta_b.py:153-157 ; run_film_tta.py:126-127 (they differ: keep both)
def delta_b_block_to_group(num_blocks: int, G: int) -> List[int]:
    per = math.ceil(num_blocks / G)
    return [min(i // per, G - 1) for i in range(num_blocks)]


def film_group_idx(num_blocks: int, G: int) -> List[int]:
    return [i * G // num_blocks for i in range(num_blocks)]


# ---- delta_experiment/scripts/early_stopping.py:166-170
def es_seed_base(video_id: str) -> int:
    return int(hashlib.md5(video_id.encode()).hexdigest()[:8], 16) % (2 ** 31)


# ---- early_stopping.py:190-243 (decision logic given the anchor losses)
def es_trace(losses: Sequence[float], check_every: int, patience: int, strategy: str, max_steps: int = 40):
    """losses[0] is the setup loss; each later entry is consumed at a check.  Returns the per-step rows
    [step, stop, best_step, checks_without_improvement] and the final (best_step, stopped_early, best_snapshot)."""
    it = iter(losses)
    best = next(it)
    best_step, cwi, stopped, snap = 0, 0, False, "init"
    rows = []
    step = 0
    try:
        while True:
            step += 1
            if step % check_every != 0:
                rows.append([step, False, None, None])
                if step >= max_steps:
                    break
                continue
            loss = next(it)
            improved = loss < best
            if improved:
                best, best_step, cwi, snap = loss, step, 0, f"snap{step}"
            else:
                cwi += 1
            stop = (cwi >= patience) if strategy == "patience" else (not improved and step > 0)
            stopped = stopped or stop
            rows.append([step, bool(stop), best_step, cwi])
            if stop or step >= max_steps:
                break
    except StopIteration:
        pass
    return rows, best_step, stopped, snap


# ---- common.py:452-470 (inputs of the conditioned loss) and :485-488 (the loss)
def build_conditioned_inputs(cond, target, sigma, eps, patch_t: int = 1, num_train_timesteps: int = 1000,
                             dtype=BF16):
    B = target.shape[0]
    T_cond, T_target = cond.shape[2], target.shape[2]
    N_cond, N_target, N_total = T_cond // patch_t, T_target // patch_t, (T_cond + T_target) // patch_t
    se = sigma.view(-1, 1, 1, 1, 1)
    noisy = (1.0 - se) * target + se * eps           # fp32 by type promotion
    hidden = torch.cat([cond, noisy], dim=2).to(dtype)
    ts = torch.zeros(B, N_total, dtype=dtype)
    ts[:, N_cond:] = (sigma * num_train_timesteps).unsqueeze(1).expand(B, N_target).to(dtype)
    return hidden, ts, N_cond


# ---- common.py:274-343 / 346-407: the unconditioned variants (same sigma on every frame, no clean prefix; the fixed form
# draws its noise from torch.Generator(device).manual_seed(42 + draw) in the LATENTS' dtype)
def build_unconditioned_inputs(latents, sigma, eps, patch_t: int = 1, num_train_timesteps: int = 1000, dtype=BF16):
    B, T = latents.shape[0], latents.shape[2]
    se = sigma.view(-1, 1, 1, 1, 1)
    noisy = ((1.0 - se) * latents + se * eps).to(dtype)
    ts = (sigma * num_train_timesteps).unsqueeze(1).expand(B, T // patch_t).to(dtype)
    return noisy, ts


def unconditioned_fixed_noise(latents, draw_idx: int) -> torch.Tensor:
    gen = torch.Generator(device=latents.device)
    gen.manual_seed(42 + draw_idx)
    return torch.randn(latents.shape, generator=gen, device=latents.device, dtype=latents.dtype)


def conditioned_loss(pred, eps, target, T_cond: int) -> torch.Tensor:
    return F.mse_loss(pred[:, :, T_cond:].to(torch.float32), (eps - target).to(torch.float32))


# ---- run_lora_tta.py:224-260 (LoRALinear) with the module's own dtype at every step
def lora_linear(x, W, b, A, B, scaling):
    orig = F.linear(x, W, b)
    return orig + F.linear(F.linear(x.to(A.dtype), A), B) * scaling


# ---- torch.nn.utils.clip_grad_norm_ + torch.optim.AdamW (foreach) on bf16 tensors, op by op
#      (driven by run_lora_tta.py:462-468, 494-497, 513-514)
def clip_coef_bf16(grads: List[torch.Tensor], max_norm: float):
    norms = torch.stack([r16(torch.linalg.vector_norm(g.float(), 2)) for g in grads])
    total = r16(torch.linalg.vector_norm(norms, 2))
    coef = r16(max_norm / r16(total + 1e-6))
    return total, torch.clamp(coef, max=1.0)


def adamw_step_bf16(p, g, m, v, step: int, lr: float, beta1=0.9, beta2=0.999, eps=1e-8, wd=0.01):
    """All tensors are fp32 holding bf16 values; returns updated (p, m, v)."""
    f = lambda x: torch.tensor(x, dtype=torch.float32)
    p = r16(p * f(1 - lr * wd))
    m = r16(m + f(1 - beta1) * (g - m))
    v = r16(v * f(beta2))
    v = r16(v + (f(1 - beta2) * g) * g)
    bc1 = 1 - beta1 ** step
    bc2 = 1 - beta2 ** step
    d = r16(v.sqrt())
    d = r16(d / f(bc2 ** 0.5))
    d = r16(d + f(eps))
    p = r16(p + f((lr / bc1) * -1) * (m / d))
    return p, m, v


def warmup_lr(lr: float, step: int, warmup_steps: int, current: float) -> float:
    """run_lora_tta.py:494-497: lr*(step+1)/warmup for step < warmup, else whatever the group already holds."""
    if step < warmup_steps and warmup_steps > 0:
        return lr * (step + 1) / warmup_steps
    return current
