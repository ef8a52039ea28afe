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


# =====================================================================================================================
# delta / FiLM / norm-tune wrappers and their optimise loops.  Pinned by tests/golden/delta_wrappers.pt, minted by
# tests/golden/make_delta_golden.py from the reference's OWN wrapper classes and loops run over oracle/dit_module.OracleDiT
# (checked in tests/test_delta_golden.py).  The functions below say WHERE each wrapper touches the DiT forward, in the
# vocabulary of `dit_oracle.dit_forward(adapters=...)`.
# =====================================================================================================================

# ---- run_delta_a.py:104, 118-126 (generation hook on t_embedder) and :168 (training forward): the same place, every frame
def delta_a_adapters(delta: torch.Tensor, depth: int, training: bool = True) -> dict:
    """delta on the t_embedder output reaches every block AND the final layer: in dit_forward that is `t_delta`."""
    return {"t_delta": delta}


# ---- run_delta_b.py:143-157 (parameters, group map), :161-165 (zero padding), :175-212 (generation hooks), :276-324 (training)
def delta_b_pad(dv: torch.Tensor, full_dim: int) -> torch.Tensor:
    return dv if dv.shape[0] >= full_dim else F.pad(dv, (0, full_dim - dv.shape[0]))


def delta_b_adapters(deltas: Sequence[torch.Tensor], delta_final: Optional[torch.Tensor], num_blocks: int, delta_target: str,
                     full_dim: int, target_blocks: str = "all", training: bool = True) -> dict:
    """Per-group deltas.  `training=True` is the wrapper's own forward (run_delta_b.py:276-324): in "hidden" mode it ALSO adds
    `delta_final` in front of the final layer; `training=False` is what `apply_to_dit()` installs for generation
    (:175-212): block hooks only — `delta_final` is trained but never applied when the video is generated."""
    G = len(deltas)
    groups = delta_b_block_to_group(num_blocks, G)
    active = parse_target_blocks(target_blocks, num_blocks)
    per_block = [delta_b_pad(deltas[groups[i]], full_dim) if (active is None or i in active) else None
                 for i in range(num_blocks)]
    if delta_target == "timestep":
        return {"block_t": per_block}
    ad = {"block_hidden": per_block}
    if training and delta_final is not None:
        ad["final_hidden"] = delta_b_pad(delta_final, full_dim)
    return ad


# ---- run_delta_c.py:117-131 (hook on the DiT output) and :159-163 (training forward)
def delta_c_adapters(delta_out: torch.Tensor) -> dict:
    return {"out_delta": delta_out}


# ---- run_film_tta.py:129-141 (partial corrections expanded to the 6C adaLN layout) and :146-163 (hooks = training forward)
def film_expand(corr: torch.Tensor, C: int, film_mode: str) -> torch.Tensor:
    if film_mode == "full":
        return corr
    z = torch.zeros(C, dtype=corr.dtype, device=corr.device)
    if film_mode == "scale_only":       # [scale_msa | scale_mlp]
        return torch.cat([z, corr[:C], z, z, corr[C:], z])
    if film_mode == "shift_scale":      # [shift_msa | scale_msa | shift_mlp | scale_mlp]
        return torch.cat([corr[:C], corr[C:2 * C], z, corr[2 * C:3 * C], corr[3 * C:], z])
    raise ValueError(f"Unknown film_mode: {film_mode}")


def film_adapters(corrections: Sequence[torch.Tensor], num_blocks: int, C: int, film_mode: str) -> dict:
    groups = film_group_idx(num_blocks, len(corrections))
    return {"film": [film_expand(corrections[groups[i]], C, film_mode) for i in range(num_blocks)]}


# ---- torch.nn.utils.clip_grad_norm_ on fp32 tensors (run_delta_a.py:267, run_delta_b.py:386-388, run_film_tta.py:305)
def clip_grads_fp32(grads: List[torch.Tensor], max_norm: float) -> List[torch.Tensor]:
    total = torch.linalg.vector_norm(torch.stack([torch.linalg.vector_norm(g, 2) for g in grads]), 2)
    coef = torch.clamp(max_norm / (total + 1e-6), max=1.0)
    return [g * coef for g in grads]


# ---- torch.optim.AdamW in fp32 as the delta scripts build it: AdamW(params, lr, betas=(0.9, 0.999), eps=1e-15) — the
#      weight decay is torch's DEFAULT 0.01 (run_delta_a.py:242, run_delta_b.py:353-356, run_delta_c.py:198, run_film_tta.py:280)
def adamw_step_fp32(p, g, m, v, step: int, lr: float, beta1=0.9, beta2=0.999, eps=1e-15, wd=0.01):
    p = p * (1 - lr * wd)
    m = m + (1 - beta1) * (g - m)
    v = v * beta2 + (1 - beta2) * g * g
    bc1, bc2 = 1 - beta1 ** step, 1 - beta2 ** step
    denom = v.sqrt() / (bc2 ** 0.5) + eps
    return p - (lr / bc1) * (m / denom), m, v


# ---- torch.optim.SGD(momentum 0, weight_decay) as run_full_tta.py:137-143 builds it
def sgd_step_fp32(p, g, lr: float, wd: float = 0.01):
    return p - lr * (g + wd * p)


def adapt_steps(loss_fn, params: List[torch.Tensor], num_steps: int, lr: float, per_param_clip: bool = False,
                max_norm: float = 1.0, optimizer: str = "adamw", eps: float = 1e-15, wd: float = 0.01, warmup_steps: int = 0):
    """The loop every optimise_* / finetune_* function of the reference shares (run_delta_a.py:256-270, run_delta_b.py:370-392,
    run_delta_c.py:212-226, run_film_tta.py:292-308, run_norm_tune_tta.py:243-260, run_full_tta.py:157-187): zero grads, optional
    linear warm-up of the LR (`lr*(step+1)/warmup`), loss, backward, clip (one global norm, or one norm PER PARAMETER for
    delta-B), optimizer step.  `loss_fn(step, params)` -> scalar.  Returns (losses, params after every step, grads of every step)."""
    ps = [p.detach().clone() for p in params]
    ms = [torch.zeros_like(p) for p in ps]
    vs = [torch.zeros_like(p) for p in ps]
    losses, trace, gtrace = [], [], []
    cur_lr = lr
    for step in range(num_steps):
        cur_lr = warmup_lr(lr, step, warmup_steps, cur_lr)
        leaves = [p.clone().requires_grad_(True) for p in ps]
        loss = loss_fn(step, leaves)
        grads = torch.autograd.grad(loss, leaves, allow_unused=True)
        live = [i for i, g in enumerate(grads) if g is not None]
        gl = [grads[i] for i in live]
        gtrace.append([None if g is None else g.detach().clone() for g in grads])
        if per_param_clip:
            gl = [clip_grads_fp32([g], max_norm)[0] for g in gl]
        else:
            gl = clip_grads_fp32(gl, max_norm)
        for i, g in zip(live, gl):
            if optimizer == "adamw":
                ps[i], ms[i], vs[i] = adamw_step_fp32(ps[i], g, ms[i], vs[i], step + 1, cur_lr, eps=eps, wd=wd)
            else:
                ps[i] = sgd_step_fp32(ps[i], g, cur_lr, wd)
        losses.append(loss.item())
        trace.append([p.clone() for p in ps])
    return losses, trace, gtrace


# ---- run_norm_tune_tta.py:74-98: which norm parameters are unfrozen, in optimizer order (names of dit_oracle.make_params)
def norm_param_names(depth: int, norm_target: str) -> List[str]:
    if norm_target not in ("cross_attn_norm", "qk_norm", "all_norm"):
        raise ValueError(f"Unknown norm_target: {norm_target}")
    names = []
    for i in range(depth):
        b = f"blocks.{i}."
        if norm_target in ("cross_attn_norm", "all_norm"):
            names += [b + "pre_crs_attn_norm.weight", b + "pre_crs_attn_norm.bias"]
        if norm_target in ("qk_norm", "all_norm"):
            names += [b + "attn.q_norm.weight", b + "attn.k_norm.weight", b + "cross_attn.q_norm.weight", b + "cross_attn.k_norm.weight"]
    return names


# ---- run_lora_tta.py:286-382: adapter order = optimizer order: per block attn.qkv, attn.proj, then cross_attn.q_linear,
#      cross_attn.kv_linear, cross_attn.proj (each when "qkv"/"proj" is targeted), then ffn.w1/w2/w3 with target_ffn
def lora_target_names(depth: int, target_modules=("qkv", "proj"), target_ffn: bool = False, target_blocks: str = "all") -> List[str]:
    active = parse_target_blocks(target_blocks, depth)
    names = []
    for i in range(depth):
        if active is not None and i not in active:
            continue
        b = f"blocks.{i}."
        if "qkv" in target_modules:
            names.append(b + "attn.qkv")
        if "proj" in target_modules:
            names.append(b + "attn.proj")
        if "qkv" in target_modules:
            names += [b + "cross_attn.q_linear", b + "cross_attn.kv_linear"]
        if "proj" in target_modules:
            names.append(b + "cross_attn.proj")
        if target_ffn:
            names += [b + "ffn.w1", b + "ffn.w2", b + "ffn.w3"]
    return names
