#!/usr/bin/env python3
"""Mint golden vectors from the REFERENCE's own delta-A/B/C, FiLM, norm-tune wrappers and the LoRA / full-model batch loops
(run in the build container only; writes tests/golden/delta_wrappers.pt + delta_wrappers.json — DATA only).

What runs is the reference's code: `DeltaAWrapper`, `DeltaBWrapper`, `DeltaCWrapper`, `FiLMAdapterWrapper`, `NormTuneForward`,
`optimize_delta_{a,b,c}`, `optimize_film_adapter`, `optimize_norm_params`, `finetune_lora_batch`,
`finetune_full_on_conditioning`, `finetune_full_batch`, `inject_lora_into_dit` and `compute_flow_matching_loss_conditioned`
from /root/reference (delta_experiment/scripts/run_delta_{a,b,c}.py, run_film_tta.py, run_norm_tune_tta.py,
lora_experiment/scripts/run_{lora,full}_tta.py), on CPU, over `oracle/dit_module.OracleDiT` — the nn.Module face of the fp32
oracle DiT with the attribute protocol those classes walk.  sigma and the noise are injected by patching
`torch.rand` / `torch.randn_like`, as tests/golden/make_golden.py does.

Per case the fixture holds: the training forward of the wrapper at a non-zero adapter state (`pred_train`), the DiT's own
forward under the wrapper's generation hooks (`pred_gen` — they differ for delta-B "hidden": run_delta_b.py:175-212 vs
:321-324), the adapter gradients of one conditioned loss there (`grads`), and a 3-step run of the reference's optimise loop
from its own initial state (`losses`, parameters after the last step, the returned dict's keys).  Structure (group maps,
parameter shapes / order) goes to the JSON file.

Re-run:  python tests/golden/make_delta_golden.py
"""
import json
import sys
from pathlib import Path

import torch
import torch.nn as nn

HERE = Path(__file__).resolve().parent
ROOT = HERE.parents[1]
sys.path.insert(0, str(HERE))
sys.path.insert(0, str(ROOT))
from _ref_loader import add_reference_paths, load_reference_module, stub_longcat  # noqa: E402

BF16 = torch.bfloat16
CFG_KW = dict(hidden_size=256, depth=4, num_heads=2, caption_channels=64)
WEIGHT_SEED, WEIGHT_STD = 77, 0.05
LR, STEPS = 1e-2, 3


def inputs():
    g = torch.Generator().manual_seed(2024)
    r = lambda *s: torch.randn(*s, generator=g)
    d = dict(cond=r(1, 16, 2, 8, 8).to(BF16), train=r(1, 16, 2, 8, 8).to(BF16), embeds=r(1, 1, 12, 64).to(BF16),
             mask=torch.tensor([[1] * 9 + [0] * 3], dtype=torch.int64),
             sig_u=torch.tensor([[0.62], [0.17], [0.88], [0.41]]), eps=torch.stack([r(1, 16, 2, 8, 8).to(BF16) for _ in range(4)]))
    # second "video" of the round-robin batch loops
    d["cond2"], d["train2"], d["embeds2"] = r(1, 16, 2, 8, 8).to(BF16), r(1, 16, 2, 8, 8).to(BF16), r(1, 1, 12, 64).to(BF16)
    d["mask2"] = torch.tensor([[1] * 5 + [0] * 7], dtype=torch.int64)
    return d


class Inject:
    """Patch torch.rand / torch.randn_like so the reference's loss draws sigma / noise number `i` of the fixture."""

    def __init__(self, I, start=0):
        self.I, self.i = I, start

    def __enter__(self):
        self._rand, self._randn_like = torch.rand, torch.randn_like
        torch.rand = lambda *a, **k: self.I["sig_u"][self.i % 4].clone()

        def randn_like(t, **k):
            e = self.I["eps"][self.i % 4].clone()
            self.i += 1
            return e
        torch.randn_like = randn_like
        return self

    def __exit__(self, *a):
        torch.rand, torch.randn_like = self._rand, self._randn_like


def fresh_dit(orc, OracleDiT):
    cfg = orc.small_config(**CFG_KW)
    P = orc.make_params(cfg, seed=WEIGHT_SEED, std=WEIGHT_STD)
    return OracleDiT(cfg, P), cfg


def fixed_inputs(O, I, k=0):
    """hidden_states / timestep of draw k, built by the (pinned) oracle restatement of common.py:452-470."""
    sigma = I["sig_u"][k] * (1.0 - 0.001) + 0.001
    return O.build_conditioned_inputs(I["cond"], I["train"], sigma, I["eps"][k])


def rnd_like(p, seed, scale=0.05):
    return torch.randn(p.shape, generator=torch.Generator().manual_seed(seed)) * scale


def main():
    stub_longcat()
    add_reference_paths()
    import common as C  # noqa
    from oracle import dit_oracle as orc, tta_oracle as O
    from oracle.dit_module import OracleDiT
    RA = load_reference_module("delta_experiment/scripts/run_delta_a.py", "ref_run_delta_a")
    RB = load_reference_module("delta_experiment/scripts/run_delta_b.py", "ref_run_delta_b")
    RC = load_reference_module("delta_experiment/scripts/run_delta_c.py", "ref_run_delta_c")
    RF = load_reference_module("delta_experiment/scripts/run_film_tta.py", "ref_run_film_tta")
    RN = load_reference_module("delta_experiment/scripts/run_norm_tune_tta.py", "ref_run_norm_tune_tta")
    RL = load_reference_module("lora_experiment/scripts/run_lora_tta.py", "ref_run_lora_tta")
    RU = load_reference_module("lora_experiment/scripts/run_full_tta.py", "ref_run_full_tta")
    I = inputs()
    T = {"inputs": I, "cfg_kw": CFG_KW, "weight_seed": WEIGHT_SEED, "weight_std": WEIGHT_STD, "lr": LR, "steps": STEPS}
    J = {"truncated_reference_files": {m.__file__.replace("/root/reference/", ""): m.__truncated_at__
                                       for m in (RA, RB, RC, RF, RN, RL, RU) if m.__truncated_at__ is not None}}
    hs, ts, ncond = fixed_inputs(O, I)
    loss_kw = dict(prompt_embeds=I["embeds"], prompt_mask=I["mask"], device="cpu", dtype=BF16)

    def wrapper_case(name, make, params_of, optimise, extra=None, persistent=False):
        """make(dit) -> wrapper; params_of(w) -> ordered adapter parameters; optimise(w) -> the reference loop's dict.
        `persistent`: the reference installs this wrapper's hooks ONCE per job and trains through them (run_film_tta.py:441)."""
        dit, cfg = fresh_dit(orc, OracleDiT)
        w = make(dit, cfg)
        ps = params_of(w)
        init = [rnd_like(p, 100 + i) for i, p in enumerate(ps)]
        with torch.no_grad():
            for p, v in zip(ps, init):
                p.copy_(v)
        w.eval()
        if persistent:
            w.apply_to_dit()
        with torch.no_grad():
            pred_train = w(hs, ts, I["embeds"], I["mask"], num_cond_latents=ncond)
            w.apply_to_dit()
            pred_gen = dit(hs, ts, I["embeds"], I["mask"], num_cond_latents=ncond)
            w.remove_from_dit()
            pred_plain = dit(hs, ts, I["embeds"], I["mask"], num_cond_latents=ncond)
        if persistent:
            w.apply_to_dit()
        w.train()
        with Inject(I):
            loss = C.compute_flow_matching_loss_conditioned(dit=w, cond_latents=I["cond"], target_latents=I["train"], **loss_kw)
        loss.backward()
        grads = [None if p.grad is None else p.grad.detach().clone() for p in ps]
        # the reference's loop from ITS initial state (zeros)
        dit2, _ = fresh_dit(orc, OracleDiT)
        w2 = make(dit2, cfg)
        if persistent:
            w2.apply_to_dit()
        with Inject(I):
            res = optimise(w2)
        case = dict(init=init, pred_train=pred_train, pred_gen=pred_gen, pred_plain=pred_plain, loss=loss.detach(), grads=grads,
                    losses=torch.tensor(res["losses"]), final=[p.detach().clone() for p in params_of(w2)])
        for k, v in res.items():
            if k not in ("losses", "es_check_time", "early_stopping_info"):
                case["ret_" + k] = torch.tensor(v) if not isinstance(v, torch.Tensor) else v
        T[name] = case
        J[name] = {"return_keys": sorted(res), "param_shapes": [list(p.shape) for p in ps]}
        if extra:
            J[name].update(extra(w))
        print(name, "loss", float(loss), "losses", res["losses"])

    opt_kw = dict(cond_latents=I["cond"], train_latents=I["train"], prompt_embeds=I["embeds"], prompt_mask=I["mask"],
                  num_steps=STEPS, lr=LR, device="cpu", dtype=BF16)

    # ------------------------------------------------------------------ delta-A (run_delta_a.py:88-305)
    wrapper_case("delta_a", lambda d, c: RA.DeltaAWrapper(d, adaln_tembed_dim=c["adaln_tembed_dim"]), lambda w: [w.delta],
                 lambda w: RA.optimize_delta_a(w, **opt_kw))

    # ------------------------------------------------------------------ delta-B (run_delta_b.py:99-421)
    def b_params(w):
        return list(w.deltas) + ([w.delta_final] if w.delta_final is not None else [])

    B_CASES = {"delta_b_t_g1": dict(num_groups=1, delta_target="timestep"),
               "delta_b_t_g3_dim32_last2": dict(num_groups=3, delta_target="timestep", delta_dim=32, target_blocks="last_2"),
               "delta_b_h_g3_dim128": dict(num_groups=3, delta_target="hidden", delta_dim=128),
               "delta_b_h_g2_full_blocks02": dict(num_groups=2, delta_target="hidden", delta_dim=256, target_blocks="0,2")}
    for name, kw in B_CASES.items():
        wrapper_case(name, lambda d, c, kw=kw: RB.DeltaBWrapper(d, adaln_tembed_dim=c["adaln_tembed_dim"], hidden_size=c["hidden_size"], **kw),
                     b_params, lambda w: RB.optimize_delta_b(w, **opt_kw),
                     extra=lambda w: {"block_to_group": list(w.block_to_group),
                                      "target_block_indices": None if w.target_block_indices is None else sorted(w.target_block_indices),
                                      "kw": kw})
    # the constructor's own error for the hidden target without --delta-dim (run_delta_b.py:149: torch.zeros(None))
    try:
        RB.DeltaBWrapper(fresh_dit(orc, OracleDiT)[0], delta_target="hidden", delta_dim=None)
        J["delta_b_hidden_without_dim"] = "ok"
    except Exception as e:  # noqa
        J["delta_b_hidden_without_dim"] = "ERR:" + type(e).__name__

    # ------------------------------------------------------------------ delta-C (run_delta_c.py:82-246)
    wrapper_case("delta_c", lambda d, c: RC.DeltaCWrapper(d, "per_channel", c["out_channels"]), lambda w: [w.delta_out],
                 lambda w: RC.optimize_delta_c(w, **opt_kw), extra=lambda w: {"mode": w.mode})
    try:
        RC.DeltaCWrapper(fresh_dit(orc, OracleDiT)[0], "full", 16)
        J["delta_c_unknown_mode"] = "ok"
    except Exception as e:  # noqa
        J["delta_c_unknown_mode"] = "ERR:" + type(e).__name__

    # ------------------------------------------------------------------ FiLM (run_film_tta.py:78-341)
    for mode, G in (("full", 2), ("shift_scale", 3), ("scale_only", 2)):
        wrapper_case(f"film_{mode}_g{G}", lambda d, c, mode=mode, G=G: RF.FiLMAdapterWrapper(d, num_groups=G, hidden_size=c["hidden_size"], film_mode=mode),
                     lambda w: list(w.corrections), lambda w: RF.optimize_film_adapter(w, **opt_kw),
                     extra=lambda w: {"group_idx": [w._get_group_idx(i) for i in range(w.num_blocks)], "correction_dim": w.correction_dim},
                     persistent=True)

    # ------------------------------------------------------------------ norm tuning (run_norm_tune_tta.py:74-283, main :371-401)
    class _NormCase(nn.Module):
        """Adapter so the shared `wrapper_case` can drive the reference's function-style norm tuning: the reference keeps the
        parameter list outside the wrapper (main(): collect_norm_params, optional delta-A vector + t_embedder hook)."""

        def __init__(self, dit, target, also_delta):
            super().__init__()
            for p in dit.parameters():
                p.requires_grad = False
            self.norm_params = RN.collect_norm_params(dit, target)
            for p in self.norm_params:
                p.requires_grad = True
            self.hook = None
            if also_delta:   # run_norm_tune_tta.py:382-390
                dp = nn.Parameter(torch.zeros(dit.config.adaln_tembed_dim))
                self.norm_params.append(dp)
                self.hook = dit.t_embedder.register_forward_hook(lambda _m, _i, out: out + dp.unsqueeze(0).to(out.dtype))
            self.inner = RN.NormTuneForward(dit)
            self.config = dit.config

        def forward(self, *a, **k):
            return self.inner(*a, **k)

        def apply_to_dit(self):
            pass

        def remove_from_dit(self):
            pass

    for target, also in (("cross_attn_norm", False), ("qk_norm", False), ("all_norm", False), ("all_norm", True)):
        name = f"norm_{target}" + ("_delta" if also else "")
        wrapper_case(name, lambda d, c, target=target, also=also: _NormCase(d, target, also), lambda w: w.norm_params,
                     lambda w: RN.optimize_norm_params(w.inner, w.norm_params, **opt_kw))

    # ------------------------------------------------------------------ index tables taken from the wrappers themselves
    class _Blocks(nn.Module):
        def __init__(self, n):
            super().__init__()
            self.blocks = nn.ModuleList([nn.Identity() for _ in range(n)])
    J["delta_b_block_to_group_48"] = {str(G): list(RB.DeltaBWrapper(_Blocks(48), num_groups=G).block_to_group) for G in (1, 2, 3, 4, 5, 7, 48)}
    J["film_group_idx_48"] = {str(G): [RF.FiLMAdapterWrapper(_Blocks(48), num_groups=G, hidden_size=8)._get_group_idx(i) for i in range(48)]
                              for G in (1, 2, 3, 4, 5, 7, 48)}

    # ------------------------------------------------------------------ LoRA batch loop (run_lora_tta.py:558-634)
    batch = [dict(cond_latents=I["cond"], train_latents=I["train"], prompt_embeds=I["embeds"], prompt_mask=I["mask"]),
             dict(cond_latents=I["cond2"], train_latents=I["train2"], prompt_embeds=I["embeds2"], prompt_mask=I["mask2"])]
    dit, cfg = fresh_dit(orc, OracleDiT)
    for p in dit.parameters():
        p.requires_grad = False
    mods = RL.inject_lora_into_dit(dit, rank=4, alpha=8.0, target_modules=["qkv", "proj"], target_ffn=False, target_blocks="all")
    with torch.no_grad():
        for i, m in enumerate(mods):   # deterministic A (the reference draws it from the global RNG); B = 0 as the reference leaves it
            m.lora_down.weight.copy_(rnd_like(m.lora_down.weight, 300 + i, 0.05))
    init = [p.detach().clone() for p in RL.get_lora_parameters(mods)]
    with Inject(I):
        res = RL.finetune_lora_batch(dit, mods, batch, num_steps=4, lr=LR, warmup_steps=3, weight_decay=0.01, max_grad_norm=1.0,
                                     device="cpu", dtype=BF16)
    T["lora_batch"] = dict(init=init, final=[p.detach().clone() for p in RL.get_lora_parameters(mods)], losses=torch.tensor(res["losses"]))
    J["lora_batch"] = {"return_keys": sorted(res), "n_modules": len(mods), "param_shapes": [list(p.shape) for p in init],
                       "hp": dict(rank=4, alpha=8.0, num_steps=4, lr=LR, warmup_steps=3, weight_decay=0.01, max_grad_norm=1.0)}
    print("lora_batch losses", res["losses"])

    # ------------------------------------------------------------------ full-model loops (run_full_tta.py:95-304)
    def summarise(dit, base):
        """Per parameter: norm of the change and its first 8 elements (the whole state would be 5 M floats per case)."""
        out = {}
        for n, p in dit.named_parameters():
            d = p.detach() - base[n]
            out[n] = torch.cat([d.norm().view(1), d.flatten()[:8]])
        return out

    for name, fn in (("full_single_sgd", lambda d: RU.finetune_full_on_conditioning(
                          d, I["cond"], I["train"], I["embeds"], I["mask"], num_steps=3, lr=1e-3, warmup_steps=2, weight_decay=0.01,
                          max_grad_norm=1.0, device="cpu", dtype=BF16, optimizer_type="sgd")),
                     ("full_batch_sgd", lambda d: RU.finetune_full_batch(d, batch, num_steps=3, lr=1e-3, warmup_steps=2, weight_decay=0.01,
                                                                        max_grad_norm=1.0, device="cpu", dtype=BF16, optimizer_type="sgd")),
                     ("full_batch_adamw", lambda d: RU.finetune_full_batch(d, batch, num_steps=3, lr=1e-4, warmup_steps=2, weight_decay=0.01,
                                                                          max_grad_norm=1.0, device="cpu", dtype=BF16, optimizer_type="adamw"))):
        dit, cfg = fresh_dit(orc, OracleDiT)
        base = {n: p.detach().clone() for n, p in dit.named_parameters()}
        with Inject(I):
            res = fn(dit)
        T[name] = dict(losses=torch.tensor(res["losses"]), change=summarise(dit, base))
        J[name] = {"return_keys": sorted(res), "n_params": len(base)}
        print(name, "losses", res["losses"])

    torch.save(T, HERE / "delta_wrappers.pt")
    (HERE / "delta_wrappers.json").write_text(json.dumps(J, indent=0))
    print("wrote delta_wrappers.pt", (HERE / "delta_wrappers.pt").stat().st_size, "bytes")


if __name__ == "__main__":
    main()
