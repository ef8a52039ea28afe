#!/usr/bin/env python3
"""Mint golden vectors from the REFERENCE's own TTA layer (run in the build container only).

The reference's `delta_experiment/scripts/{common,early_stopping}.py` and
`lora_experiment/scripts/run_lora_tta.py` import cleanly once the un-vendored `longcat_video.*` module names are
stubbed; this script imports them from /root/reference, drives them on CPU with injected
sigma / noise and a deterministic toy DiT, and writes inputs + expected outputs to tests/golden/*.json|*.pt.
Only DATA is written — no reference source text.  Re-run:  python tests/golden/make_golden.py
"""
import hashlib
import json
import math
import sys
import types
from pathlib import Path

import torch
import torch.nn as nn

REF = Path("/root/reference")
OUT = Path(__file__).resolve().parent
sys.path.insert(0, str(OUT))


def _stub_longcat():
    names = ["longcat_video", "longcat_video.modules", "longcat_video.modules.scheduling_flow_match_euler_discrete",
             "longcat_video.modules.autoencoder_kl_wan", "longcat_video.modules.longcat_video_dit",
             "longcat_video.pipeline_longcat_video", "longcat_video.modules.lora_utils",
             "longcat_video.context_parallel", "longcat_video.context_parallel.context_parallel_util"]
    for n in names:
        sys.modules[n] = types.ModuleType(n)
    sys.modules["longcat_video.modules.scheduling_flow_match_euler_discrete"].FlowMatchEulerDiscreteScheduler = object
    sys.modules["longcat_video.modules.autoencoder_kl_wan"].AutoencoderKLWan = object
    sys.modules["longcat_video.modules.longcat_video_dit"].LongCatVideoTransformer3DModel = object
    sys.modules["longcat_video.pipeline_longcat_video"].LongCatVideoPipeline = object
    sys.modules["longcat_video.pipeline_longcat_video"].retrieve_latents = lambda x: x
    sys.modules["longcat_video.modules.lora_utils"].LoRAModule = object


class ToyDiT(nn.Module):
    """Deterministic stand-in with the attributes the reference's loss / LoRA code touches.  Linear in its
    input so results are reproducible to fp32 rounding on any backend:
        pred = a * hidden + b * (timestep/1000 broadcast over frames) + attn.qkv-path contribution."""

    def __init__(self, dim=32, n_blocks=3, dtype=torch.bfloat16):
        super().__init__()
        self.config = types.SimpleNamespace(patch_size=(1, 2, 2), adaln_tembed_dim=8, hidden_size=dim, out_channels=16)
        self.patch_size = (1, 2, 2)
        self.seen = []

        class Attn(nn.Module):
            def __init__(s):
                super().__init__()
                s.qkv = nn.Linear(dim, 3 * dim)
                s.proj = nn.Linear(dim, dim)

        class XAttn(nn.Module):
            def __init__(s):
                super().__init__()
                s.q_linear = nn.Linear(dim, dim)
                s.kv_linear = nn.Linear(dim, 2 * dim)
                s.proj = nn.Linear(dim, dim)

        class FFN(nn.Module):
            def __init__(s):
                super().__init__()
                s.w1 = nn.Linear(dim, 2 * dim, bias=False)
                s.w2 = nn.Linear(2 * dim, dim, bias=False)
                s.w3 = nn.Linear(dim, 2 * dim, bias=False)

        class Block(nn.Module):
            def __init__(s):
                super().__init__()
                s.attn, s.cross_attn, s.ffn = Attn(), XAttn(), FFN()

        self.blocks = nn.ModuleList([Block() for _ in range(n_blocks)])
        self.inp = nn.Linear(16, dim)
        self.out = nn.Linear(dim, 16)
        g = torch.Generator().manual_seed(5)
        for p in self.parameters():
            p.data = (torch.randn(p.shape, generator=g) * 0.2)
        self.to(dtype)

    def forward(self, hidden_states, timestep, encoder_hidden_states=None, encoder_attention_mask=None,
                num_cond_latents=0):
        self.seen.append({"hidden_states": hidden_states.detach().clone(), "timestep": timestep.detach().clone(),
                          "num_cond_latents": int(num_cond_latents)})
        B, C, T, H, W = hidden_states.shape
        x = hidden_states.permute(0, 2, 3, 4, 1)  # [B,T,H,W,C]
        h = self.inp(x) + (timestep.to(x.dtype) / 1000.0).view(B, T, 1, 1, 1)
        for blk in self.blocks:
            q = blk.attn.qkv(h)[..., : h.shape[-1]]
            h = h + blk.attn.proj(torch.tanh(q))
            h = h + blk.cross_attn.proj(blk.cross_attn.q_linear(h)) * 0.1
            h = h + blk.ffn.w2(torch.nn.functional.silu(blk.ffn.w1(h)) * blk.ffn.w3(h)) * 0.1
        return self.out(h).permute(0, 4, 1, 2, 3).to(torch.float32)


def t2l(t):
    return t.detach().to(torch.float32).flatten().tolist()


def main():
    _stub_longcat()
    sys.path.insert(0, str(REF / "delta_experiment" / "scripts"))
    sys.path.insert(0, str(REF / "lora_experiment" / "scripts"))
    import common as C  # noqa
    import early_stopping as ES  # noqa
    import run_lora_tta as L  # noqa
    gold = {}

    # ---------------------------------------------------------------- (a4) latent split + budget mirror
    rows = []
    for T in range(1, 60):
        for n_ctx in (0, 1, 2, 3, 4, 5, 8, 100):
            for frac in (0.25, 0.1, 0.5):
                lat = torch.arange(T, dtype=torch.float32).view(1, 1, T, 1, 1)
                try:
                    c, tr, v = C.split_tta_latents(lat, n_ctx, frac)
                    rows.append([T, n_ctx, frac, t2l(c), t2l(tr), None if v is None else t2l(v)])
                except Exception as e:  # T=1: recorded as an error row
                    rows.append([T, n_ctx, frac, "ERR", type(e).__name__, None])
    gold["split_tta_latents"] = rows
    pairs = [(14, 14), (32, 14), (28, 14), (48, 14), (49, 14), (120, 14), (200, 14), (2, 2), (7, 7), (24, 24), (1, 1), (5, 9)]
    gold["estimate_tta_split_budget"] = [[a, b, C.estimate_tta_split_budget(a, b)] for a, b in pairs]
    gold["estimate_latent_len"] = [[n, C._estimate_latent_len(n)] for n in range(-2, 130)]
    # num_frames_valid arithmetic of generate_video_continuation (common.py:589-593)
    gold["num_frames_valid"] = [[n, ((n - 1 + 4 - 1) // 4) * 4 + 1] for n in range(1, 130)]

    # ---------------------------------------------------------------- (a6) target-block parsing, group maps
    cases = ["all", "ALL ", "last_1", "last_4", "last_48", "0,5,10", " 3 , 47", "last_0", "last_49", "48", "-1", "x"]
    pt = []
    for c in cases:
        try:
            r = L._parse_target_blocks(c, 48)
            pt.append([c, None if r is None else sorted(r)])
        except Exception as e:
            pt.append([c, "ERR:" + type(e).__name__])
    gold["parse_target_blocks"] = pt
    # the two group maps are attributes of the reference's wrappers: taken from them (run_delta_b.py:153-157, run_film_tta.py:126-127)
    from _ref_loader import load_reference_module
    RB = load_reference_module("delta_experiment/scripts/run_delta_b.py", "ref_run_delta_b")
    RF = load_reference_module("delta_experiment/scripts/run_film_tta.py", "ref_run_film_tta")

    class _Blocks(nn.Module):
        def __init__(self, n):
            super().__init__()
            self.blocks = nn.ModuleList([nn.Identity() for _ in range(n)])
    gold["delta_b_block_to_group"] = {str(G): list(RB.DeltaBWrapper(_Blocks(48), num_groups=G).block_to_group) for G in (1, 2, 3, 4, 5, 7, 48)}
    gold["film_group_idx"] = {str(G): [RF.FiLMAdapterWrapper(_Blocks(48), num_groups=G, hidden_size=8)._get_group_idx(i) for i in range(48)]
                              for G in (1, 2, 3, 4, 5, 7, 48)}

    # ---------------------------------------------------------------- (a11) early-stopper seeds + decision traces
    vids = ["v_ApplyEyeMakeup_g01_c01", "", "panda/000123.mp4", "a" * 40]
    gold["es_seed_base"] = [[v, int(hashlib.md5(v.encode()).hexdigest()[:8], 16) % (2 ** 31)] for v in vids]
    traces = []
    seqs = {
        "improve_then_rise": [1.0, 0.9, 0.8, 0.85, 0.86, 0.87, 0.5],
        "flat": [1.0, 1.0, 1.0, 1.0, 1.0],
        "first_rise": [1.0, 0.9, 0.95, 0.1],
        "always_better": [1.0, 0.9, 0.8, 0.7, 0.6, 0.5],
        "nan_mid": [1.0, float("nan"), 0.5, 0.6, 0.7, 0.8],
    }
    for strategy in ("patience", "first_rise"):
        for name, seq in seqs.items():
            for check_every, patience in ((5, 3), (1, 2), (2, 1)):
                es = ES.AnchoredEarlyStopper(check_every=check_every, patience=patience, strategy=strategy)
                it = iter(seq)
                es._compute_anchor_loss = lambda it=it: next(it)
                es.model = object()
                es.best_state = "init"
                es.best_loss = es._compute_anchor_loss()
                es.loss_history.append((0, es.best_loss))
                out = []
                step = 0
                try:
                    while True:
                        step += 1
                        stop, info = es.step(step, save_fn=lambda s=step: f"snap{s}")
                        out.append([step, bool(stop), info.get("best_step"), info.get("checks_without_improvement")])
                        if stop or step >= 40:
                            break
                except StopIteration:
                    pass
                traces.append({"strategy": strategy, "seq": name, "losses": [None if x != x else x for x in seq],
                               "check_every": check_every, "patience": patience, "steps": out,
                               "best_state": es.best_state, "best_step": es.best_step, "stopped_early": es.stopped_early,
                               "history_len": len(es.loss_history)})
    gold["es_traces"] = traces

    (OUT / "tta_index.json").write_text(json.dumps(gold, indent=0))

    # ---------------------------------------------------------------- (a1/a2) conditioned loss with injected sigma / eps
    tens = {}
    torch.manual_seed(0)
    dit = ToyDiT().eval()
    cond = torch.randn(1, 16, 2, 4, 6).to(torch.bfloat16)
    tgt = torch.randn(1, 16, 3, 4, 6).to(torch.bfloat16)
    sig_u = torch.tensor([0.37109375])  # the torch.rand draw; sigma = u*(max-min)+min
    eps = torch.randn(1, 16, 3, 4, 6).to(torch.bfloat16)
    real_rand, real_randn_like = torch.rand, torch.randn_like
    torch.rand = lambda *a, **k: sig_u.clone()
    torch.randn_like = lambda t, **k: eps.clone()
    try:
        loss = C.compute_flow_matching_loss_conditioned(dit, cond, tgt, None, None, device="cpu", dtype=torch.bfloat16)
        seen = dit.seen[-1]
        tens["loss_cond"] = dict(cond=cond, target=tgt, sig_u=sig_u, eps=eps, hidden_states=seen["hidden_states"],
                                 timestep=seen["timestep"], num_cond_latents=torch.tensor(seen["num_cond_latents"]),
                                 loss=loss.detach())
        # empty conditioning edge case (split 0/1/0)
        cond0 = torch.zeros(1, 16, 0, 4, 6, dtype=torch.bfloat16)
        loss0 = C.compute_flow_matching_loss_conditioned(dit, cond0, tgt, None, None, device="cpu", dtype=torch.bfloat16)
        seen = dit.seen[-1]
        tens["loss_cond_empty"] = dict(target=tgt, sig_u=sig_u, eps=eps, hidden_states=seen["hidden_states"],
                                       timestep=seen["timestep"],
                                       num_cond_latents=torch.tensor(seen["num_cond_latents"]), loss=loss0.detach())
    finally:
        torch.rand, torch.randn_like = real_rand, real_randn_like
    noises = [torch.randn(1, 16, 3, 4, 6).to(torch.bfloat16) for _ in range(2)]
    lf = C.compute_flow_matching_loss_conditioned_fixed(dit, cond, tgt, None, None, [0.25, 0.5, 0.75], noises,
                                                        device="cpu", dtype=torch.bfloat16)
    tens["loss_cond_fixed"] = dict(cond=cond, target=tgt, noises=torch.stack(noises), sigmas=torch.tensor([0.25, 0.5, 0.75]),
                                   loss=torch.tensor(lf),
                                   timesteps=torch.stack([s["timestep"] for s in dit.seen[-6:]]))
    tens["toy_dit_state"] = {k: v.clone() for k, v in dit.state_dict().items()}

    # ---------------------------------------------------------------- (a5) LoRALinear forward / backward
    for dt_name, dt in (("bf16", torch.bfloat16), ("fp32", torch.float32)):
        torch.manual_seed(1)
        base = nn.Linear(64, 96)
        lora = L.LoRALinear(base, rank=4, alpha=16.0).to(dt)
        with torch.no_grad():
            lora.lora_up.weight.copy_((torch.randn(96, 4) * 0.1).to(dt))  # non-zero B so both grads are exercised
        x = (torch.randn(2, 10, 64)).to(dt).requires_grad_(True)
        y = lora(x)
        gy = torch.randn(2, 10, 96).to(dt)
        y.backward(gy)
        tens[f"lora_linear_{dt_name}"] = dict(
            W=base.weight.detach().clone(), b=base.bias.detach().clone(), A=lora.lora_down.weight.detach().clone(),
            B=lora.lora_up.weight.detach().clone(), x=x.detach().clone(), y=y.detach().clone(), gy=gy,
            dx=x.grad.clone(), dA=lora.lora_down.weight.grad.clone(), dB=lora.lora_up.weight.grad.clone(),
            scaling=torch.tensor(lora.scaling))

    # ---------------------------------------------------------------- (a6/a8/a9) injection order + the whole inner loop
    torch.manual_seed(2)
    dit = ToyDiT(dtype=torch.bfloat16)
    for p in dit.parameters():
        p.requires_grad = False
    mods = L.inject_lora_into_dit(dit, rank=2, alpha=4.0, target_modules=["qkv", "proj"], target_ffn=True,
                                  target_blocks="last_2")
    order = []
    for name, m in dit.named_modules():
        if isinstance(m, L.LoRALinear):
            order.append(name)
    inj = {"named_modules_order": order, "n_modules": len(mods)}
    g = torch.Generator().manual_seed(3)
    with torch.no_grad():
        for m in mods:  # deterministic re-init (reset_lora_weights draws from the global RNG)
            m.lora_down.weight.copy_((torch.rand(m.lora_down.weight.shape, generator=g) - 0.5).to(torch.bfloat16) * 0.3)
            m.lora_up.weight.zero_()
    init = [p.detach().clone() for p in L.get_lora_parameters(mods)]
    n_steps = 5
    us = [torch.tensor([u]) for u in (0.11, 0.52, 0.93, 0.27, 0.66)]
    es_list = [torch.randn(1, 16, 3, 4, 6, generator=g).to(torch.bfloat16) for _ in range(n_steps)]
    cnt = {"i": 0}

    def fake_rand(*a, **k):
        return us[cnt["i"]].clone()

    def fake_randn_like(t, **k):
        e = es_list[cnt["i"]].clone()
        cnt["i"] += 1
        return e

    torch.rand, torch.randn_like = fake_rand, fake_randn_like
    try:
        res = L.finetune_lora_on_conditioning(dit, mods, cond, tgt, None, None, num_steps=n_steps, lr=2e-2,
                                              warmup_steps=3, weight_decay=0.01, max_grad_norm=1.0, device="cpu",
                                              dtype=torch.bfloat16, early_stopper=None)
    finally:
        torch.rand, torch.randn_like = real_rand, real_randn_like
    final = [p.detach().clone() for p in L.get_lora_parameters(mods)]
    tens["inner_loop"] = dict(cond=cond, target=tgt, sig_u=torch.stack(us), eps=torch.stack(es_list),
                              losses=torch.tensor(res["losses"]), init_params=init, final_params=final,
                              base_state={k: v.clone() for k, v in dit.state_dict().items() if "lora_" not in k},
                              hp=dict(lr=2e-2, warmup_steps=3, weight_decay=0.01, max_grad_norm=1.0, rank=2, alpha=4.0,
                                      num_steps=n_steps))
    inj["param_shapes"] = [list(p.shape) for p in init]
    inj["state_keys"] = [k for k in dit.state_dict().keys() if "lora_" in k]
    (OUT / "lora_injection.json").write_text(json.dumps(inj, indent=0))

    # ---------------------------------------------------------------- AdamW + clip + warm-up on bf16 tensors (op-level trace)
    torch.manual_seed(4)
    ps = [nn.Parameter((torch.randn(7, 33) * 0.5).to(torch.bfloat16)), nn.Parameter((torch.randn(130) * 0.5).to(torch.bfloat16))]
    opt = torch.optim.AdamW(ps, lr=2e-3, betas=(0.9, 0.999), weight_decay=0.01, eps=1e-8)
    trace = {"init": [p.detach().clone() for p in ps], "grads": [], "after": [], "norms": [], "lrs": []}
    for step in range(6):
        if step < 3:
            for pg in opt.param_groups:
                pg["lr"] = 2e-3 * (step + 1) / 3
        gs = [(torch.randn(p.shape) * (3.0 if step % 2 == 0 else 0.05)).to(torch.bfloat16) for p in ps]
        for p, gq in zip(ps, gs):
            p.grad = gq.clone()
        n = torch.nn.utils.clip_grad_norm_(ps, 1.0)
        opt.step()
        trace["grads"].append(gs)
        trace["norms"].append(n.detach().clone())
        trace["after"].append([p.detach().clone() for p in ps])
        trace["lrs"].append(opt.param_groups[0]["lr"])
    tens["adamw_trace"] = trace

    torch.save(tens, OUT / "tta_tensors.pt")
    print("wrote", [p.name for p in OUT.iterdir()])


if __name__ == "__main__":
    main()
