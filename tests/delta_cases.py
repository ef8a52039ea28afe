"""Shared by tests/test_delta_golden.py (CPU: oracle vs fixtures) and tests/test_gpu_delta_golden.py (GPU: product vs fixtures):
the fixtures minted from the reference's own delta / FiLM / norm-tune wrappers (tests/golden/make_delta_golden.py), the oracle
weights they were minted over, and the mapping from a case's adapter parameters to `dit_oracle.dit_forward` arguments."""
import json
from pathlib import Path

import torch

from oracle import dit_oracle as orc, tta_oracle as O

G = Path(__file__).resolve().parent / "golden"
T = torch.load(G / "delta_wrappers.pt")
J = json.loads((G / "delta_wrappers.json").read_text())
I = T["inputs"]
CFG = orc.small_config(**T["cfg_kw"])
P = {k: v.float() for k, v in orc.make_params(CFG, seed=T["weight_seed"], std=T["weight_std"]).items()}
DEPTH, C, CT = CFG["depth"], CFG["hidden_size"], CFG["adaln_tembed_dim"]


def _inputs(k, video=0):
    cond, train, emb, mask = ((I["cond"], I["train"], I["embeds"], I["mask"]) if video == 0 else
                              (I["cond2"], I["train2"], I["embeds2"], I["mask2"]))
    sigma = I["sig_u"][k % 4] * (1.0 - 0.001) + 0.001
    hs, ts, ncond = O.build_conditioned_inputs(cond, train, sigma, I["eps"][k % 4])
    return hs, ts, ncond, emb, mask, train, I["eps"][k % 4]


def _loss(k, params=None, video=0, bf16=False, **fw):
    hs, ts, ncond, emb, mask, train, eps = _inputs(k, video)
    pred = orc.dit_forward(params or P, CFG, hs, ts, emb, mask, ncond, bf16=bf16, **fw)
    return O.conditioned_loss(pred, eps, train, ncond), pred


def _forward_kw(name, ps, training=True):
    """Adapter parameters (in the reference's order) -> keyword arguments of dit_oracle.dit_forward."""
    if name == "delta_a":
        return dict(t_delta=ps[0])
    if name.startswith("delta_b"):
        kw = J[name]["kw"]
        hidden = kw["delta_target"] == "hidden"
        deltas, dfin = (ps[:-1], ps[-1]) if hidden else (ps, None)
        return dict(adapters=O.delta_b_adapters(deltas, dfin, DEPTH, kw["delta_target"], C if hidden else CT,
                                                kw.get("target_blocks", "all"), training=training))
    if name == "delta_c":
        return dict(adapters=O.delta_c_adapters(ps[0]))
    if name.startswith("film_"):
        mode = name[len("film_"):name.rfind("_g")]
        return dict(adapters=O.film_adapters(ps, DEPTH, C, mode))
    raise KeyError(name)


WRAPPER_CASES = [k for k in T if k.startswith(("delta_", "film_"))]
NORM_CASES = [k for k in T if k.startswith("norm_")]


def norm_case(name):
    """-> (norm_target, also_tune_delta, oracle parameter names in the reference's optimizer order)"""
    also = name.endswith("_delta")
    target = name[len("norm_"):-len("_delta")] if also else name[len("norm_"):]
    return target, also, O.norm_param_names(DEPTH, target)


def norm_forward_kw(name, ps):
    _, also, names = norm_case(name)
    Q = dict(P)
    Q.update(dict(zip(names, ps)))
    return dict(params=Q, **(dict(t_delta=ps[-1]) if also else {}))
