"""Full-width oracle checks of the LoRA-only backward and of the inner step (round 3, VERDICT r2 "Next round" #1).

The reference's step is `loss -> backward -> clip_grad_norm_ -> AdamW.step` over LoRA adapters injected into a frozen
13.6 B-parameter DiT (lora_experiment/scripts/run_lora_tta.py:502-515, LoRALinear :224-260).  Rounds 1-2 checked the
whole-model adapter gradients against oracle autograd only on a 256-wide, 2-head toy.  Here the SAME comparison runs at
the real width - hidden 4096, 32 heads of 128, FFN 11 008, text width 4096, adaLN 512, rank 8, alpha 16 - where the fused
extra-K LoRA tile, `tn_skinny` at K = 4096 / 12 288 / 11 008, the packed-qkv gradient, the fused-SwiGLU training path and
the residual-fork `dres` are composed for the first time:

  (i)   depth 2, K1 tokens (1 280 = 5 x 16 x 16), num_cond_latents in {0, 2}, adapters on qkv + proj (and once with the FFN
        adapters too), non-zero A and B: the loss and EVERY adapter gradient vs torch autograd over `oracle/dit_oracle.py`
        evaluated in fp32 on the card with the adapters folded in as W + s B A;
  (ii)  the same at the reference's operating point: 480p, 3 context + 1 target latent frame = 6 240 tokens;
  (iii) a depth sweep 2 / 8 / 48 at K1 tokens: max / median adapter-gradient rel-L2 printed next to the oracle's OWN
        bf16-vs-fp32 gradient gap (the oracle re-run with its bf16 rounding points, whose autograd rounds the gradients to
        bf16 at the same points);
  (v)   (round 4) one step at BASELINE.json config 3's own size: 720p, 4 context + 3 target latent frames = 25 200 tokens;
  (iv)  three full inner steps (fused clip + AdamW, warm-up) vs the fp32 oracle + `clip_grad_norm_` + `torch.optim.AdamW` on
        bf16 adapter tensors: per-step losses, and the adapter weights after step 3.

The oracle runs with plain fp32 torch ops on the GPU (`tests/test_gpu_denoise_parity.py::test_oracle_is_device_independent`
shows that this changes nothing beyond fp32 summation order: 1e-5).  Measured values are written to
gpurun_out/backward_parity.json and quoted in DESIGN.md §3.  THE criterion of (i), (ii), (iii) and (v) is relative (round 4): the HIP
gradients are no further from the fp32 oracle's than the oracle's own bf16-rounding-point evaluation is
(`hip_vs_fp32 < 1.5 x oracle_bf16_vs_fp32 + 1e-3`, max and median over the adapters); the absolute bounds beside it (at most
1.5 x a round-3 measurement) stay as a drift alarm."""
import json
import os
from pathlib import Path

import pytest
import torch

from conftest import rel_l2

pytestmark = pytest.mark.gpu
BF16 = torch.bfloat16
DEV = "cuda"
_REPORT = {}

ATTN_NAMES = ("attn.qkv", "attn.proj", "cross_attn.q_linear", "cross_attn.kv_linear", "cross_attn.proj")
FFN_NAMES = ("ffn.w1", "ffn.w2", "ffn.w3")


def _record(key, value):
    _REPORT[key] = value
    out = Path(os.environ.get("GRAFT_REPO_ROOT", Path(__file__).resolve().parents[1])) / "gpurun_out"
    try:
        out.mkdir(exist_ok=True)
        (out / "backward_parity.json").write_text(json.dumps(_REPORT, indent=1))
    except OSError:
        pass


def _cfg(depth):
    from oracle import dit_oracle as D
    cfg = D.small_config(hidden_size=4096, depth=depth, num_heads=32, caption_channels=4096)
    cfg["adaln_tembed_dim"] = 512
    assert cfg["ffn_hidden"] == 11008
    return cfg


def _model(depth, seed=1234):
    from longcat_video.modules.longcat_video_dit import LongCatVideoTransformer3DModel
    m = LongCatVideoTransformer3DModel(device=DEV, dtype=BF16, depth=depth).init_synthetic_(seed)
    for p in m.parameters():
        p.requires_grad = False
    return m


def _inject(m, ffn=False, seed=5, rank=8, alpha=16.0):
    """Adapters as the reference injects them (kaiming-uniform A) but with a NON-ZERO B (what it is after a few steps)."""
    from tta.lora import inject_lora_into_dit
    mods = inject_lora_into_dit(m, rank=rank, alpha=alpha, target_modules=["qkv", "proj"], target_ffn=ffn)
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for lm in mods:
            bound = 1.0 / (lm.lora_down.weight.shape[1] ** 0.5)          # kaiming_uniform_(a = sqrt 5) on [r, in]
            lm.lora_down.weight.copy_(((torch.rand(lm.lora_down.weight.shape, generator=g) * 2 - 1) * bound).to(BF16))
            lm.lora_up.weight.copy_((torch.randn(lm.lora_up.weight.shape, generator=g) * 0.02).to(BF16))
    return mods


def _adapter_names(depth, ffn):
    per = ATTN_NAMES + (FFN_NAMES if ffn else ())
    return [f"blocks.{i}.{n}" for i in range(depth) for n in per]


def _inputs(T, h, w, ncond, sigma_t, L=512, valid=77, seed=30):
    g = torch.Generator().manual_seed(seed)
    hs = torch.randn(1, 16, T, h, w, generator=g).to(BF16)
    y = torch.randn(1, 1, L, 4096, generator=g).to(BF16)
    mask = torch.zeros(1, L, dtype=torch.int64); mask[:, :valid] = 1
    ts = torch.zeros(1, T); ts[:, ncond:] = sigma_t
    eps = torch.randn(1, 16, T - ncond, h, w, generator=g).to(BF16)
    x0 = torch.randn(1, 16, T - ncond, h, w, generator=g).to(BF16)
    return tuple(t.to(DEV) for t in (hs, ts.to(BF16), y, mask, eps, x0))


def _base_params(m):
    """fp32 copies of the frozen weights under the ORACLE's names (the injected model stores `<name>.original.<param>`)."""
    return {k.replace(".original.", "."): v.detach().float() for k, v in m.state_dict().items()
            if ".lora_down." not in k and ".lora_up." not in k}


def _oracle_grads(P32, cfg, mods, names, inp, ncond, bf16=False, leaves=None):
    """loss and d loss / d (A, B) of every adapter from torch autograd over the oracle with W + s * B @ A folded in."""
    from oracle import dit_oracle as D
    hs, ts, y, mask, eps, x0 = inp
    P2 = dict(P32)
    own = leaves is None
    leaves = [] if own else leaves
    it = iter(leaves)
    for n, lm in zip(names, mods):
        if own:
            A_ = lm.lora_down.weight.detach().float().requires_grad_(True)
            B_ = lm.lora_up.weight.detach().float().requires_grad_(True)
            leaves += [A_, B_]
        else:
            A_, B_ = next(it), next(it)
        P2[n + ".weight"] = P32[n + ".weight"] + lm.scaling * (B_.float() @ A_.float())
    pred = D.dit_forward(P2, cfg, hs, ts, y, mask, ncond, bf16=bf16)
    loss = torch.nn.functional.mse_loss(pred[:, :, ncond:].float(), (eps - x0).float())
    loss.backward()
    return loss.detach(), leaves


def _hip_grads(m, mods, inp, ncond):
    from tta.flow_matching import fm_mse_loss
    from tta.lora import get_lora_parameters
    hs, ts, y, mask, eps, x0 = inp
    params = get_lora_parameters(mods)
    for p in params:
        p.grad = None
    m.train()
    pred = m(hs, ts, y, mask, num_cond_latents=ncond)
    loss = fm_mse_loss(pred, eps, x0, ncond)
    loss.backward()
    m.eval()
    return loss.detach(), params


def _compare(tag, m, mods, names, cfg, inp, ncond, depth_used=None):
    P32 = _base_params(m)
    loss, params = _hip_grads(m, mods, inp, ncond)
    ref_loss, leaves = _oracle_grads(P32, cfg, mods, names, inp, ncond, bf16=False)
    n_act = len(names) * 2
    errs = [rel_l2(p.grad, l.grad) for p, l in zip(params[:n_act], leaves)]
    per_kind = {}
    for i, e in enumerate(errs):
        kind = names[i // 2].split(".", 2)[2] + (".A" if i % 2 == 0 else ".B")
        per_kind[kind] = max(per_kind.get(kind, 0.0), e)
    srt = sorted(errs)
    row = {"tokens": int(inp[0].shape[2] * inp[0].shape[3] * inp[0].shape[4] // 4), "ncond": ncond, "adapters": len(names),
           "loss_hip": float(loss), "loss_oracle_fp32": float(ref_loss), "loss_rel": abs(float(loss) - float(ref_loss)) / abs(float(ref_loss)),
           "grad_rel_l2_max": srt[-1], "grad_rel_l2_median": srt[len(srt) // 2], "grad_rel_l2_worst_by_kind": per_kind}
    print(f"{tag}: loss {row['loss_hip']:.6f} vs {row['loss_oracle_fp32']:.6f} (rel {row['loss_rel']:.1e}); adapter gradients rel-L2 "
          f"max {srt[-1]:.2e} median {srt[len(srt) // 2]:.2e}; worst by kind " + ", ".join(f"{k} {v:.1e}" for k, v in per_kind.items()))
    _record(tag, row)
    assert all(p.grad is not None and torch.isfinite(p.grad.float()).all() for p in params[:n_act])
    return row, P32, leaves, params


def _assert_relative(tag, row, P32, cfg, mods, names, inp, ncond, leaves, params):
    """THE criterion (the absolute bounds in the tests are a secondary drift alarm): every adapter gradient of the HIP path is no
    further from the fp32 oracle's than the oracle's OWN bf16-rounding-point evaluation is (max and median over the adapters)."""
    g32 = [l.grad.clone() for l in leaves]
    for l in leaves:
        l.grad = None
    _oracle_grads(P32, cfg, mods, names, inp, ncond, bf16=True, leaves=leaves)
    own = sorted(rel_l2(l.grad, g) for l, g in zip(leaves, g32))
    hip_vs_bf = sorted(rel_l2(p.grad, l.grad) for p, l in zip(params, leaves))
    row.update(oracle_bf16_vs_fp32_max=own[-1], oracle_bf16_vs_fp32_median=own[len(own) // 2],
               hip_vs_oracle_bf16_max=hip_vs_bf[-1], hip_vs_oracle_bf16_median=hip_vs_bf[len(hip_vs_bf) // 2])
    _record(tag, row)
    print(f"{tag}: oracle bf16 vs fp32 max {own[-1]:.2e} / median {own[len(own) // 2]:.2e};  HIP vs oracle bf16 max {hip_vs_bf[-1]:.2e}")
    assert row["grad_rel_l2_max"] < 1.5 * row["oracle_bf16_vs_fp32_max"] + 1e-3, (tag, row)
    assert row["grad_rel_l2_median"] < 1.5 * row["oracle_bf16_vs_fp32_median"] + 1e-3, (tag, row)


@pytest.fixture(scope="module")
def dit2_lora():
    m = _model(2)
    mods = _inject(m, ffn=False)
    yield m, mods
    del m
    torch.cuda.empty_cache()


# ------------------------------------------------------------------------------------------------ (i) K1 tokens
@pytest.mark.parametrize("ncond", [0, 2])
def test_full_width_lora_gradients_k1_tokens(dit2_lora, ncond):
    m, mods = dit2_lora
    cfg = _cfg(2)
    inp = _inputs(5, 32, 32, ncond, 431.0)
    row, P32, leaves, params = _compare(f"i_k1_depth2_ncond{ncond}", m, mods, _adapter_names(2, False), cfg, inp, ncond)
    _assert_relative(f"i_k1_depth2_ncond{ncond}", row, P32, cfg, mods, _adapter_names(2, False), inp, ncond, leaves, params)
    # secondary drift alarm, measured (round 3): loss 8.6e-5 / 8.5e-6 relative, gradients max 1.82e-2 / 1.69e-2 (cross_attn.q_linear.A), median 4.2e-3
    assert row["loss_rel"] < 4.4e-4
    assert row["grad_rel_l2_max"] < 2.7e-2 and row["grad_rel_l2_median"] < 6.3e-3


def test_full_width_lora_gradients_k1_tokens_with_ffn_adapters():
    """`--target-ffn`: adapters on w1 / w2 / w3 as well (K = 4096 -> 11 008 -> 4096); the FFN then takes the unfused path."""
    m = _model(2, seed=77)
    mods = _inject(m, ffn=True, seed=6)
    cfg = _cfg(2)
    inp = _inputs(5, 32, 32, 2, 612.0, seed=31)
    row, P32, leaves, params = _compare("i_k1_depth2_ncond2_ffn", m, mods, _adapter_names(2, True), cfg, inp, 2)
    _assert_relative("i_k1_depth2_ncond2_ffn", row, P32, cfg, mods, _adapter_names(2, True), inp, 2, leaves, params)
    assert row["loss_rel"] < 4.4e-4                                    # measured 5.7e-5; gradients max 7.5e-3, median 3.0e-3
    assert row["grad_rel_l2_max"] < 1.1e-2 and row["grad_rel_l2_median"] < 4.5e-3


# ------------------------------------------------------------------------------------------------ (ii) 6 240 tokens
def test_full_width_lora_gradients_reference_operating_point(dit2_lora):
    """480p, tta_total = num_cond = 14 frames -> 3 context + 1 target latent frame = 6 240 tokens: the shape behind the
    reference's published 4.25 s per step."""
    m, mods = dit2_lora
    cfg = _cfg(2)
    inp = _inputs(4, 60, 104, 3, 777.0, seed=32)
    row, P32, leaves, params = _compare("ii_480p_6240tok_depth2_ncond3", m, mods, _adapter_names(2, False), cfg, inp, 3)
    assert row["tokens"] == 6240
    _assert_relative("ii_480p_6240tok_depth2_ncond3", row, P32, cfg, mods, _adapter_names(2, False), inp, 3, leaves, params)
    assert row["loss_rel"] < 4.4e-4                                    # drift alarm; measured 2.9e-4; gradients max 7.9e-3, median 3.9e-3
    assert row["grad_rel_l2_max"] < 1.2e-2 and row["grad_rel_l2_median"] < 5.9e-3


def test_full_width_lora_gradients_config3_720p_25200_tokens(dit2_lora):
    """BASELINE.json config 3 (49x720p + LoRA TTA; tta_total 32 frames -> 4 context + 3 target latent frames at 720p = 25 200
    tokens, SURVEY §8 row a1 / §8(d) "K3-TTA"): ONE full-width depth-2 LoRA step's loss and every adapter gradient vs oracle
    autograd at the size the bench times (round 4; before, 25 200 tokens were covered kernel by kernel only).  The fp32 oracle
    keeps its softmax matrices for the backward: 32 heads x (14 400^2 + 10 800 x 25 200) x 4 B = 61 GB per block."""
    m, mods = dit2_lora
    cfg = _cfg(2)
    inp = _inputs(7, 90, 160, 4, 655.0, seed=34)
    names = _adapter_names(2, False)
    row, P32, leaves, params = _compare("v_720p_25200tok_depth2_ncond4", m, mods, names, cfg, inp, 4)
    assert row["tokens"] == 25200
    torch.cuda.empty_cache()
    _assert_relative("v_720p_25200tok_depth2_ncond4", row, P32, cfg, mods, names, inp, 4, leaves, params)
    assert row["loss_rel"] < 1e-3


# ------------------------------------------------------------------------------------------------ (iii) depth sweep
def test_full_width_lora_gradient_depth_sweep():
    """Depth 2 / 8 / 48 at K1 tokens.  Beside every HIP-vs-fp32 number: the oracle's own bf16-vs-fp32 gradient gap (same
    folded adapters, bf16 rounding points in the forward, gradients rounded to bf16 at the same points by autograd)."""
    m = _model(48)
    mods = _inject(m, ffn=False, seed=9)
    blocks = m.blocks
    inp = _inputs(5, 32, 32, 2, 431.0, seed=33)
    rows = {}
    try:
        for depth in (2, 8, 48):
            m.blocks = blocks[:depth]
            cfg = _cfg(depth)
            names = _adapter_names(depth, False)
            act = mods[:len(names)]
            row, P32, leaves, params = _compare(f"iii_k1_depth{depth}", m, act, names, cfg, inp, 2)
            g32 = [l.grad.clone() for l in leaves]
            for l in leaves:
                l.grad = None
            _oracle_grads(P32, cfg, act, names, inp, 2, bf16=True, leaves=leaves)
            own = sorted(rel_l2(l.grad, g) for l, g in zip(leaves, g32))
            hip_vs_bf = sorted(rel_l2(p.grad, l.grad) for p, l in zip(params, leaves))
            row.update(oracle_bf16_vs_fp32_max=own[-1], oracle_bf16_vs_fp32_median=own[len(own) // 2],
                       hip_vs_oracle_bf16_max=hip_vs_bf[-1], hip_vs_oracle_bf16_median=hip_vs_bf[len(hip_vs_bf) // 2])
            rows[depth] = row
            print(f"depth {depth:2d}: HIP vs fp32 oracle max {row['grad_rel_l2_max']:.2e} / median {row['grad_rel_l2_median']:.2e};  "
                  f"oracle bf16 vs fp32 max {own[-1]:.2e} / median {own[len(own) // 2]:.2e};  HIP vs oracle bf16 max {hip_vs_bf[-1]:.2e}")
            _record(f"iii_k1_depth{depth}", row)
            del P32, leaves, g32
            torch.cuda.empty_cache()
    finally:
        m.blocks = blocks
    for depth, r in rows.items():
        # the HIP backward sits no further from the fp32 truth than ~the oracle's own bf16 evaluation does
        # measured (HIP / oracle-bf16, max | median): depth 2 9.1e-3 / 8.8e-3 | 4.0e-3 / 3.4e-3; depth 8 1.35e-2 / 1.27e-2 | 4.8e-3 /
        # 4.4e-3; depth 48 2.49e-2 / 2.54e-2 | 1.03e-2 / 1.01e-2; loss 2.0e-4 ... 5.5e-4 relative
        assert r["grad_rel_l2_max"] < 1.5 * r["oracle_bf16_vs_fp32_max"] + 1e-3, (depth, r)
        assert r["grad_rel_l2_median"] < 1.5 * r["oracle_bf16_vs_fp32_median"] + 1e-3, (depth, r)
        assert r["loss_rel"] < 8.3e-4, (depth, r)


# ------------------------------------------------------------------------------------------------ (iv) three inner steps
def test_three_full_width_inner_steps_match_oracle_adamw():
    """`finetune_lora_on_conditioning` (fused LoRA GEMMs, fused clip + AdamW, warm-up) for 3 steps at full width against
    the fp32 oracle + `clip_grad_norm_` + `torch.optim.AdamW` on bf16 adapter tensors (the reference's optimizer state
    dtype, run_lora_tta.py:332, 462-468), with the same injected sigma / noise."""
    from tta.inner_loop import finetune_lora_on_conditioning
    from tta.lora import get_lora_parameters
    m = _model(2, seed=4321)
    mods = _inject(m, ffn=False, seed=12)
    cfg = _cfg(2)
    names = _adapter_names(2, False)
    g = torch.Generator().manual_seed(40)
    cond = torch.randn(1, 16, 2, 32, 32, generator=g).to(BF16).to(DEV)
    target = torch.randn(1, 16, 3, 32, 32, generator=g).to(BF16).to(DEV)
    pe = torch.randn(1, 1, 512, 4096, generator=g).to(BF16).to(DEV)
    pm = torch.zeros(1, 512, dtype=torch.int64, device=DEV); pm[:, :77] = 1
    steps, lr, warm, wd, clip = 3, 2e-4, 3, 0.01, 1.0
    sig_u = [torch.rand(1, generator=g) for _ in range(steps)]
    noise = [torch.randn(target.shape, generator=g).to(BF16) for _ in range(steps)]
    params = get_lora_parameters(mods)
    init = [p.detach().clone() for p in params]
    P32 = _base_params(m)

    # ---- the oracle's run: bf16 leaves, fp32 forward with the folded weights, torch's own clip + AdamW
    from oracle import dit_oracle as D
    ref = [torch.nn.Parameter(p.clone()) for p in init]
    ropt = torch.optim.AdamW(ref, lr=lr, betas=(0.9, 0.999), weight_decay=wd, eps=1e-8)
    ref_losses, ref_norms = [], []
    for step in range(steps):
        ropt.zero_grad(set_to_none=True)
        if step < warm:
            for pg in ropt.param_groups:
                pg["lr"] = lr * (step + 1) / warm
        sigma = (sig_u[step] * (1.0 - 0.001) + 0.001).to(DEV)
        sx = sigma.view(1, 1, 1, 1, 1)
        noisy = ((1.0 - sx) * target.float() + sx * noise[step].to(DEV).float()).to(BF16)        # common.py:458-466
        hs = torch.cat([cond, noisy], dim=2)
        ts = torch.zeros(1, 5, device=DEV, dtype=BF16)
        ts[:, 2:] = (sigma * 1000).unsqueeze(1).expand(1, 3).to(BF16)
        P2 = dict(P32)
        for i, (n, lm) in enumerate(zip(names, mods)):
            P2[n + ".weight"] = P32[n + ".weight"] + lm.scaling * (ref[2 * i + 1].float() @ ref[2 * i].float())
        pred = D.dit_forward(P2, cfg, hs, ts, pe, pm, 2, bf16=False)
        loss = torch.nn.functional.mse_loss(pred[:, :, 2:].float(), (noise[step].to(DEV) - target).float())
        loss.backward()
        ref_norms.append(float(torch.nn.utils.clip_grad_norm_(ref, clip)))
        ropt.step()
        ref_losses.append(float(loss.detach()))
        del P2, pred, loss

    # ---- the product's run, same sigma / noise draws injected
    cnt = {"i": 0}
    real_rand, real_randn_like = torch.rand, torch.randn_like
    torch.rand = lambda *a, **k: sig_u[cnt["i"]].to(DEV).clone()

    def fake_randn_like(x, **k):
        e = noise[cnt["i"]].to(DEV).clone(); cnt["i"] += 1
        return e
    torch.randn_like = fake_randn_like
    try:
        res = finetune_lora_on_conditioning(m, mods, cond, target, pe, pm, num_steps=steps, lr=lr, warmup_steps=warm,
                                            weight_decay=wd, max_grad_norm=clip, device=DEV, dtype=BF16, early_stopper=None)
    finally:
        torch.rand, torch.randn_like = real_rand, real_randn_like
    loss_rel = [abs(a - b) / abs(b) for a, b in zip(res["losses"], ref_losses)]
    # adapter weights after step 3: distance in bf16 ulps of the value, and the rel-L2 of the accumulated UPDATE
    within1 = within2 = total = 0
    upd_err = []
    for p, r, p0 in zip(params, ref, init):
        a, b = p.detach().float(), r.detach().float()
        ulp = torch.maximum(a.abs(), b.abs()) * 2.0 ** -7 + 1e-30
        d = (a - b).abs()
        within1 += int((d <= ulp).sum()); within2 += int((d <= 2 * ulp).sum()); total += d.numel()
        upd_err.append(rel_l2(a - p0.float(), b - p0.float()))
    w_err = [rel_l2(p.detach(), r.detach()) for p, r in zip(params, ref)]
    row = {"losses_hip": res["losses"], "losses_oracle": ref_losses, "loss_rel": loss_rel, "oracle_grad_norms": ref_norms,
           "weights_rel_l2_max": max(w_err), "update_rel_l2_max": max(upd_err), "update_rel_l2_median": sorted(upd_err)[len(upd_err) // 2],
           "frac_within_1ulp": within1 / total, "frac_within_2ulp": within2 / total}
    print(f"3 inner steps: losses {['%.5f' % x for x in res['losses']]} vs oracle {['%.5f' % x for x in ref_losses]}; "
          f"adapter weights rel-L2 max {row['weights_rel_l2_max']:.2e}; update rel-L2 max {row['update_rel_l2_max']:.2e} "
          f"median {row['update_rel_l2_median']:.2e}; within 1 ulp {row['frac_within_1ulp']:.4f}, 2 ulp {row['frac_within_2ulp']:.4f}")
    _record("iv_three_inner_steps_k1_depth2", row)
    # measured (round 3): losses within 1.0e-4 / 1.7e-4 / 1.7e-4 relative; adapter weights rel-L2 max 1.12e-3; 98.73 % of the
    # 852 k adapter elements within one bf16 ulp, 99.46 % within two.  Adam's first steps move every element by ~lr * sign(g):
    # an element whose (tiny) gradient changes sign between the two implementations ends up to 2 lr apart - the rest of the
    # distribution, and why the rel-L2 of the accumulated UPDATE (3.1e-2 median, 5.9e-2 max) is larger than that of the weights.
    assert max(loss_rel) < 2.6e-4
    assert row["weights_rel_l2_max"] < 1.7e-3 and row["update_rel_l2_max"] < 8.8e-2 and row["update_rel_l2_median"] < 4.6e-2
    assert row["frac_within_1ulp"] > 0.981 and row["frac_within_2ulp"] > 0.9919
