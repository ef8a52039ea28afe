"""CPU: pin the oracle (oracle/tta_oracle.py) and the product's pure-host logic (tta.latent_split, tta.lora parsing,
tta.early_stopping decisions) against golden vectors minted from the reference itself (tests/golden/make_golden.py)."""
import json
from pathlib import Path

import pytest
import torch

G = Path(__file__).resolve().parent / "golden"
IDX = json.loads((G / "tta_index.json").read_text())
TENS = torch.load(G / "tta_tensors.pt")
BF16 = torch.bfloat16


def test_split_matches_reference_bit_exact():
    from oracle import tta_oracle as O
    from tta.latent_split import split_tta_latents
    n = 0
    for T, n_ctx, frac, c, tr, v in IDX["split_tta_latents"]:
        if c == "ERR":
            continue
        tc, tt, tv = O.split_sizes(T, n_ctx, frac)
        assert (tc, tt, tv) == (len(c), len(tr), 0 if v is None else len(v)), (T, n_ctx, frac)
        lat = torch.arange(T, dtype=torch.float32).view(1, 1, T, 1, 1)
        pc, pt, pv = split_tta_latents(lat, n_ctx, frac)
        assert pc.flatten().tolist() == c and pt.flatten().tolist() == tr
        assert (pv is None and v is None) or pv.flatten().tolist() == v
        assert pc.is_contiguous() and pt.is_contiguous()
        n += 1
    assert n > 1000


def test_budget_latent_len_and_frame_rounding():
    from oracle import tta_oracle as O
    from tta import latent_split as S
    for a, b, exp in IDX["estimate_tta_split_budget"]:
        assert S.estimate_tta_split_budget(a, b) == exp
        tc, tt, tv = O.split_sizes(O.latent_len(a), O.latent_len(b))
        assert (tc, tt, tv) == (exp["cond_latents"], exp["train_latents"], exp["val_latents"])
    for n, exp in IDX["estimate_latent_len"]:
        assert S._estimate_latent_len(n) == exp == O.latent_len(n)
    for n, exp in IDX["num_frames_valid"]:
        assert S.num_frames_valid(n) == exp == O.num_frames_valid(n)


def test_parse_target_blocks_and_group_maps():
    from oracle import tta_oracle as O
    from tta.lora import _parse_target_blocks
    for spec, exp in IDX["parse_target_blocks"]:
        for fn in (O.parse_target_blocks, _parse_target_blocks):
            if isinstance(exp, str) and exp.startswith("ERR"):
                with pytest.raises(Exception):
                    fn(spec, 48)
            else:
                got = fn(spec, 48)
                assert (got is None and exp is None) or sorted(got) == exp
    for Gs, exp in IDX["delta_b_block_to_group"].items():
        assert O.delta_b_block_to_group(48, int(Gs)) == exp
    for Gs, exp in IDX["film_group_idx"].items():
        assert O.film_group_idx(48, int(Gs)) == exp


def test_es_seed_and_decision_traces():
    from oracle import tta_oracle as O
    from tta.early_stopping import AnchoredEarlyStopper, es_seed_base
    for vid, exp in IDX["es_seed_base"]:
        assert O.es_seed_base(vid) == exp == es_seed_base(vid)
    for tr in IDX["es_traces"]:
        losses = [float("nan") if x is None else x for x in tr["losses"]]
        rows, best_step, stopped, snap = O.es_trace(losses, tr["check_every"], tr["patience"], tr["strategy"])
        exp_rows = tr["steps"]
        checks = [r for r in rows if r[2] is not None or r[1]]
        exp_checks = [r for r in exp_rows if r[2] is not None]
        assert [r[:2] for r in rows] == [r[:2] for r in exp_rows], tr
        assert checks == exp_checks
        assert (best_step, stopped, snap) == (tr["best_step"], tr["stopped_early"], tr["best_state"])
        # the product class, driven with the same fake anchor losses
        es = AnchoredEarlyStopper(check_every=tr["check_every"], patience=tr["patience"], strategy=tr["strategy"])
        it = iter(losses)
        es._compute_anchor_loss = lambda it=it: next(it)
        es.model, es.best_state = object(), "init"
        es.best_loss = es._compute_anchor_loss()
        es.loss_history.append((0, es.best_loss))
        out, step = [], 0
        try:
            while True:
                step += 1
                stop, info = es.step(step, save_fn=lambda s=step: f"snap{s}")
                out.append([step, bool(stop), info.get("best_step"), info.get("checks_without_improvement")])
                if stop or step >= 40:
                    break
        except StopIteration:
            pass
        assert out == exp_rows
        assert (es.best_state, es.best_step, es.stopped_early, len(es.loss_history)) == \
               (tr["best_state"], tr["best_step"], tr["stopped_early"], tr["history_len"])


def test_conditioned_loss_inputs_bit_exact_and_loss_value():
    from oracle import tta_oracle as O
    for key in ("loss_cond", "loss_cond_empty"):
        t = TENS[key]
        cond = t.get("cond", torch.zeros(1, 16, 0, 4, 6, dtype=BF16))
        sigma = t["sig_u"] * (1.0 - 0.001) + 0.001
        hs, ts, ncond = O.build_conditioned_inputs(cond, t["target"], sigma, t["eps"])
        assert torch.equal(hs, t["hidden_states"]) and torch.equal(ts, t["timestep"])
        assert ncond == int(t["num_cond_latents"])
    # the sigma*1000 -> bf16 round trip is visible in the fixture
    t = TENS["loss_cond"]
    s1000 = float((t["sig_u"] * 0.999 + 0.001) * 1000)
    assert float(t["timestep"][0, -1]) != s1000 and abs(float(t["timestep"][0, -1]) - s1000) < 2.0
    f = TENS["loss_cond_fixed"]
    assert torch.equal(f["timesteps"][0][0, -1], (torch.tensor(0.25) * 1000).to(BF16))


def test_lora_linear_forward_backward():
    from oracle import tta_oracle as O
    for key, tol in (("lora_linear_fp32", 1e-5), ("lora_linear_bf16", 0.0)):
        t = TENS[key]
        x = t["x"].clone().requires_grad_(True)
        A = t["A"].clone().requires_grad_(True)
        B = t["B"].clone().requires_grad_(True)
        y = O.lora_linear(x, t["W"], t["b"], A, B, float(t["scaling"]))
        y.backward(t["gy"])
        for got, exp in ((y, t["y"]), (x.grad, t["dx"]), (A.grad, t["dA"]), (B.grad, t["dB"])):
            if tol == 0.0:
                assert torch.equal(got.detach(), exp)
            else:
                assert torch.allclose(got.detach(), exp, rtol=tol, atol=tol)


def test_adamw_clip_trace_bit_exact():
    """The op-by-op bf16 emulation reproduces torch's clip_grad_norm_ + AdamW(foreach) on bf16 tensors exactly."""
    from oracle import tta_oracle as O
    tr = TENS["adamw_trace"]
    ps = [p.float() for p in tr["init"]]
    ms = [torch.zeros_like(p) for p in ps]
    vs = [torch.zeros_like(p) for p in ps]
    lr_cur = 2e-3
    for step in range(len(tr["grads"])):
        lr_cur = O.warmup_lr(2e-3, step, 3, lr_cur)
        assert abs(lr_cur - tr["lrs"][step]) < 1e-15
        grads = [g.float() for g in tr["grads"][step]]
        total, coef = O.clip_coef_bf16(grads, 1.0)
        assert torch.equal(total.to(BF16), tr["norms"][step].to(BF16))
        for i in range(len(ps)):
            g = O.r16(grads[i] * coef)
            ps[i], ms[i], vs[i] = O.adamw_step_bf16(ps[i], g, ms[i], vs[i], step + 1, lr_cur)
            assert torch.equal(ps[i].to(BF16), tr["after"][step][i]), (step, i)


def test_unconditioned_loss_inputs_bit_exact():
    """the unconditioned losses (common.py:274-407; imported by every runner, called by none) — what the
    reference's functions handed to the toy DiT, restated by the oracle bit for bit."""
    from oracle import tta_oracle as O
    U = torch.load(G / "uncond_loss.pt")
    t = U["uncond"]
    sigma = t["sig_u"] * (1.0 - 0.001) + 0.001
    hs, ts = O.build_unconditioned_inputs(t["latents"], sigma, t["eps"])
    assert torch.equal(hs, t["hidden_states"]) and torch.equal(ts, t["timestep"]) and t["num_cond_latents"] == 0
    f = U["uncond_fixed"]
    k = 0
    for s in f["sigmas"]:
        for d in range(f["noise_draws"]):
            eps = O.unconditioned_fixed_noise(f["latents"], d)
            hs, ts = O.build_unconditioned_inputs(f["latents"], torch.tensor([s]), eps)
            assert torch.equal(hs, f["hidden_states"][k]) and torch.equal(ts, f["timesteps"][k]), (s, d)
            k += 1
    assert k == f["hidden_states"].shape[0] == 4
