"""CPU: the oracle's restatement of the delta-A/B/C, FiLM, norm-tune wrappers and of the LoRA / full-model batch loops against
fixtures minted from the REFERENCE's own classes and loops (tests/golden/make_delta_golden.py -> delta_wrappers.pt/json).

Everything here is fp32 on both sides, so the bounds are summation-order noise: predictions 2e-5, gradients 2e-4 relative,
parameters after three Adam steps 2e-3 relative (Adam's first steps are sign-like: a 1e-7 gradient difference at a
near-zero gradient element moves that element by lr), losses 1e-5.  Structure (group maps, shapes, dict keys) is exact.
"""
import pytest
import torch

from conftest import rel_l2
from oracle import dit_oracle as orc, tta_oracle as O
from oracle.dit_module import OracleDiT

from delta_cases import (C, CFG, CT, DEPTH, I, J, NORM_CASES, P, T, WRAPPER_CASES, _forward_kw, _inputs, _loss, norm_case,
                         norm_forward_kw)


def test_structure_tables_come_from_the_reference_wrappers():
    for Gs, exp in J["delta_b_block_to_group_48"].items():
        assert O.delta_b_block_to_group(48, int(Gs)) == exp
    for Gs, exp in J["film_group_idx_48"].items():
        assert O.film_group_idx(48, int(Gs)) == exp
    for name in WRAPPER_CASES:
        if name.startswith("delta_b"):
            assert O.delta_b_block_to_group(DEPTH, J[name]["kw"]["num_groups"]) == J[name]["block_to_group"]
            act = O.parse_target_blocks(J[name]["kw"].get("target_blocks", "all"), DEPTH)
            assert (None if act is None else sorted(act)) == J[name]["target_block_indices"]
        if name.startswith("film_"):
            assert O.film_group_idx(DEPTH, len(T[name]["init"])) == J[name]["group_idx"]
    # what the reference's constructors do with arguments they cannot serve
    assert J["delta_b_hidden_without_dim"] == "ERR:TypeError" and J["delta_c_unknown_mode"] == "ERR:ValueError"
    assert sorted(J["truncated_reference_files"]) == ["delta_experiment/scripts/run_delta_a.py", "delta_experiment/scripts/run_delta_c.py",
                                                     "delta_experiment/scripts/run_film_tta.py", "delta_experiment/scripts/run_norm_tune_tta.py"]


@pytest.mark.parametrize("name", WRAPPER_CASES)
def test_wrapper_forward_gradients_and_three_steps(name):
    case = T[name]
    hs, ts, ncond, emb, mask, _, _ = _inputs(0)
    init = [p.clone() for p in case["init"]]
    with torch.no_grad():
        plain = orc.dit_forward(P, CFG, hs, ts, emb, mask, ncond, bf16=False)
        train = orc.dit_forward(P, CFG, hs, ts, emb, mask, ncond, bf16=False, **_forward_kw(name, init, True))
        gen = orc.dit_forward(P, CFG, hs, ts, emb, mask, ncond, bf16=False, **_forward_kw(name, init, False))
    assert rel_l2(plain, case["pred_plain"]) < 2e-5
    assert rel_l2(train, case["pred_train"]) < 2e-5
    assert rel_l2(gen, case["pred_gen"]) < 2e-5
    if "delta_b_h" in name:   # the generation hooks leave delta_final out: the two forwards of the reference differ
        assert rel_l2(case["pred_gen"], case["pred_train"]) > 1e-3
    else:
        assert rel_l2(case["pred_gen"], case["pred_train"]) < 2e-5
    leaves = [p.clone().requires_grad_(True) for p in init]
    loss, _ = _loss(0, **_forward_kw(name, leaves, True))
    assert abs(loss.item() - float(case["loss"])) < 1e-5 * abs(float(case["loss"]))
    grads = torch.autograd.grad(loss, leaves, allow_unused=True)
    for g, e in zip(grads, case["grads"]):
        assert (g is None) == (e is None)
        if e is not None:
            assert rel_l2(g, e) < 2e-4
    zeros = [torch.zeros_like(p) for p in init]
    losses, trace, _ = O.adapt_steps(lambda step, ps: _loss(step, **_forward_kw(name, ps, True))[0], zeros, T["steps"], T["lr"],
                                     per_param_clip=name.startswith("delta_b"))
    assert torch.allclose(torch.tensor(losses), case["losses"], rtol=1e-5)
    for p, e in zip(trace[-1], case["final"]):
        assert rel_l2(p, e) < 2e-3, name
    keys = {"delta_a": ["delta_norm"], "delta_c": ["delta_out_norm", "delta_out_values"]}.get(
        name, ["delta_norms"] if name.startswith("delta_b") else ["correction_norm"])
    assert set(J[name]["return_keys"]) >= set(keys) | {"losses", "early_stopping_info"}
    if name.startswith("delta_b"):   # the final-layer delta's norm is reported too (run_delta_b.py:413-415)
        assert len(case["ret_delta_norms"]) == len(case["final"])


@pytest.mark.parametrize("name", NORM_CASES)
def test_norm_tuning_gradients_and_three_steps(name):
    case = T[name]
    target, also, names = norm_case(name)
    assert [list(P[n].shape) for n in names] + ([[CT]] if also else []) == J[name]["param_shapes"]

    def fw(ps):
        return norm_forward_kw(name, ps)

    hs, ts, ncond, emb, mask, _, _ = _inputs(0)
    init = [p.clone() for p in case["init"]]
    with torch.no_grad():
        kw = fw(init)
        pred = orc.dit_forward(kw.pop("params"), CFG, hs, ts, emb, mask, ncond, bf16=False, **kw)
    assert rel_l2(pred, case["pred_train"]) < 2e-5 and rel_l2(pred, case["pred_gen"]) < 2e-5
    leaves = [p.clone().requires_grad_(True) for p in init]
    loss, _ = _loss(0, **fw(leaves))
    for g, e in zip(torch.autograd.grad(loss, leaves), case["grads"]):
        assert rel_l2(g, e) < 2e-4
    start = [P[n].clone() for n in names] + ([torch.zeros(CT)] if also else [])
    losses, trace, _ = O.adapt_steps(lambda step, ps: _loss(step, **fw(ps))[0], start, T["steps"], T["lr"])
    assert torch.allclose(torch.tensor(losses), case["losses"], rtol=1e-5)
    for p, e in zip(trace[-1], case["final"]):
        assert rel_l2(p, e) < 2e-3
    assert J[name]["return_keys"] == ["early_stopping_info", "losses"]


def test_lora_batch_loop_round_robin():
    """finetune_lora_batch (run_lora_tta.py:558-634): video `step % n`, LR warm-up, one global clip, AdamW(eps 1e-8, wd 0.01)."""
    case, hp = T["lora_batch"], J["lora_batch"]["hp"]
    targets = O.lora_target_names(DEPTH)
    assert len(targets) == J["lora_batch"]["n_modules"]
    assert [list(p.shape) for p in case["init"]] == J["lora_batch"]["param_shapes"]
    s = hp["alpha"] / hp["rank"]

    def loss_fn(step, ps):
        Q = dict(P)
        for i, n in enumerate(targets):   # [down, up] per module: W + s * up @ down
            Q[n + ".weight"] = P[n + ".weight"] + s * ps[2 * i + 1] @ ps[2 * i]
        return _loss(step, params=Q, video=step % 2)[0]

    losses, trace, _ = O.adapt_steps(loss_fn, [p.clone() for p in case["init"]], hp["num_steps"], hp["lr"], eps=1e-8,
                                     wd=hp["weight_decay"], warmup_steps=hp["warmup_steps"], max_norm=hp["max_grad_norm"])
    assert torch.allclose(torch.tensor(losses), case["losses"], rtol=2e-5)
    assert max(rel_l2(p, e) for p, e in zip(trace[-1], case["final"])) < 2e-3
    assert J["lora_batch"]["return_keys"] == ["early_stopping_info", "es_check_time", "losses", "train_time"]


@pytest.mark.parametrize("name,opt,lr,batch", [("full_single_sgd", "sgd", 1e-3, False), ("full_batch_sgd", "sgd", 1e-3, True),
                                               ("full_batch_adamw", "adamw", 1e-4, True)])
def test_full_model_loops(name, opt, lr, batch):
    """finetune_full_on_conditioning / finetune_full_batch (run_full_tta.py:95-304): every parameter trains; SGD(momentum 0,
    wd 0.01) or AdamW(eps 1e-8); warm-up 2; one global clip at 1.0."""
    case = T[name]
    names = list(OracleDiT(CFG).state_dict().keys())       # nn.Module.parameters() order = the optimizer's order
    assert len(names) == J[name]["n_params"]

    def loss_fn(step, ps):
        return _loss(step, params=dict(zip(names, ps)), video=(step % 2 if batch else 0))[0]

    losses, trace, _ = O.adapt_steps(loss_fn, [P[n].clone() for n in names], 3, lr, optimizer=opt, eps=1e-8, wd=0.01, warmup_steps=2)
    assert torch.allclose(torch.tensor(losses), case["losses"], rtol=2e-5)
    worst = 0.0
    for n, p in zip(names, trace[-1]):
        d = p - P[n]
        got = torch.cat([d.norm().view(1), d.flatten()[:8]])
        worst = max(worst, rel_l2(got, case["change"][n]))
    assert worst < (5e-3 if opt == "adamw" else 1e-3), worst
    assert J[name]["return_keys"] == ["early_stopping_info", "es_check_time", "losses", "train_time"]
