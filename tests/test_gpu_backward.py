"""GPU parity of the backward kernels and the TTA inner loop.

References: torch autograd over the CPU oracle functions in fp32 (no bf16 rounding: roundings are straight-through
in both), and the golden vectors minted from the reference's own LoRALinear / AdamW / inner loop.
Tolerances: gradients are bf16 tensors -> relative L2 <= 1e-2 against the fp32 autograd reference (inputs are bf16,
P/dS are rounded to bf16 before their MFMAs exactly like the forward); optimizer results bit-exact up to the stated
fraction of 1-ulp flips caused by the norm's summation order.
"""
import json
import sys
from pathlib import Path

import pytest
import torch

from conftest import rel_l2

pytestmark = pytest.mark.gpu
BF16 = torch.bfloat16
DEV = "cuda"
G = Path(__file__).resolve().parent / "golden"


def _randn(*shape, seed=0, scale=1.0, dtype=BF16):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(*shape, generator=g) * scale).to(dtype)


@pytest.mark.parametrize("B,H,Nq,Nk", [(1, 2, 200, 200), (2, 1, 70, 333), (1, 1, 256, 128), (1, 2, 33, 5),
                                       (1, 2, 4100, 77), (2, 3, 2500, 128)])      # the last two: query sweep split over workgroups
def test_attention_backward(B, H, Nq, Nk):
    from lcv_hip import ops
    D = 128
    q = _randn(B, Nq, H, D, seed=1); k = _randn(B, Nk, H, D, seed=2); v = _randn(B, Nk, H, D, seed=3)
    do = _randn(B, Nq, H, D, seed=4)
    scale = D ** -0.5
    qd, kd, vd, dod = q.to(DEV), k.to(DEV), v.to(DEV), do.to(DEV)
    o, lse = ops.attention(qd, kd, vd, scale, need_lse=True)
    dq = torch.empty_like(qd); dk = torch.empty_like(kd); dv = torch.empty_like(vd)
    ops.attention_bwd(qd, kd, vd, o, dod, lse, dq, dk, dv, scale)
    qf, kf, vf = (t.float().permute(0, 2, 1, 3).requires_grad_(True) for t in (q, k, v))
    s = (qf @ kf.transpose(-1, -2)) * scale
    ref = torch.softmax(s, -1) @ vf
    ref.backward(do.float().permute(0, 2, 1, 3))
    assert rel_l2(dq.permute(0, 2, 1, 3), qf.grad, bound=4.0e-3) < 4.0e-3
    assert rel_l2(dk.permute(0, 2, 1, 3), kf.grad, bound=3.8e-3) < 3.8e-3
    assert rel_l2(dv.permute(0, 2, 1, 3), vf.grad, bound=3.6e-3) < 3.6e-3
    # ... and against the checker that rounds where the kernels round (P and dS as bf16 MFMA operands, bf16 results)
    sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
    from oracle import dit_oracle as orc
    rq, rk, rv = orc.sdpa_backward_at_kernel_rounding(*(t.permute(0, 2, 1, 3) for t in (q, k, v, do)), scale, scale_inside=True,
                                                      o=o.cpu().permute(0, 2, 1, 3), lse=lse.cpu())
    assert rel_l2(dq.permute(0, 2, 1, 3), rq, bound=1.6e-4) < 1.6e-4
    assert rel_l2(dk.permute(0, 2, 1, 3), rk, bound=2.7e-4) < 2.7e-4
    assert rel_l2(dv.permute(0, 2, 1, 3), rv, bound=2.0e-4) < 2.0e-4
    # no atomics on any path (the split query sweep of the short-key form keeps a slice per split): a second call gives the same bits
    dq_b = torch.empty_like(qd); dk_b = torch.empty_like(kd); dv_b = torch.empty_like(vd)
    ops.attention_bwd(qd, kd, vd, o, dod, lse, dq_b, dk_b, dv_b, scale)
    assert torch.equal(dq, dq_b) and torch.equal(dk, dk_b) and torch.equal(dv, dv_b)
    # accumulate_kv adds into existing dk/dv
    dk2 = dk.clone(); dv2 = dv.clone()
    ops.attention_bwd(qd, kd, vd, o, dod, lse, dq, dk2, dv2, scale, accumulate_kv=True)
    assert rel_l2(dk2, 2 * dk.float(), bound=1e-2) < 1e-2 and rel_l2(dv2, 2 * dv.float(), bound=1e-2) < 1e-2


def test_attention_backward_fuzz_against_the_kernels_rounding_points():
    """20 seeded random shapes through both families of backward kernels (general scale; unit scale = q in log2 units, the DiT's
    self-attention) against `sdpa_backward_at_kernel_rounding` fed the forward's own O and lse: P and dS as bf16 MFMA operands with
    the scale on the side of the rounding each family has it.  What is left is fp32 summation order."""
    import math
    from lcv_hip import ops
    sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
    from oracle import dit_oracle as orc
    D = 128
    g = torch.Generator().manual_seed(78)
    ri = lambda lo, hi: int(torch.randint(lo, hi + 1, (1,), generator=g))
    worst = 0.0
    for case in range(20):
        B, H = ri(1, 2), ri(1, 2)
        Nq = [ri(1, 600), 32 * ri(1, 10) + ri(-1, 1)][case % 2]
        Nk = [ri(1, 900), 128 * ri(1, 5) + ri(-1, 1), ri(1, 100)][(case // 2) % 3]
        unit = case % 4 >= 2
        q = _randn(B, Nq, H, D, seed=4000 + case); k = _randn(B, Nk, H, D, seed=5000 + case); v = _randn(B, Nk, H, D, seed=6000 + case)
        do = _randn(B, Nq, H, D, seed=7000 + case)
        scale = D ** -0.5
        if unit:
            q = (q.float() * (scale * math.log2(math.e))).to(BF16)
            scale = math.log(2.0)
        qd, kd, vd, dod = q.to(DEV), k.to(DEV), v.to(DEV), do.to(DEV)
        o, lse = ops.attention(qd, kd, vd, scale, need_lse=True)
        dq = torch.empty_like(qd); dk = torch.empty_like(kd); dv = torch.empty_like(vd)
        ops.attention_bwd(qd, kd, vd, o, dod, lse, dq, dk, dv, scale)
        rq, rk, rv = orc.sdpa_backward_at_kernel_rounding(*(t.permute(0, 2, 1, 3) for t in (q, k, v, do)), scale, scale_inside=not unit,
                                                          o=o.cpu().permute(0, 2, 1, 3), lse=lse.cpu())
        e = max(rel_l2(dq.permute(0, 2, 1, 3), rq), rel_l2(dk.permute(0, 2, 1, 3), rk), rel_l2(dv.permute(0, 2, 1, 3), rv))
        worst = max(worst, e)
        assert e < 5.0e-4, (case, B, H, Nq, Nk, unit, e)
    print(f"attention backward fuzz: worst rel-L2 against the rounding-point checker {worst:.2e}")


@pytest.mark.parametrize("B,H,Nq,Nk", [(1, 2, 300, 300), (2, 2, 70, 333), (1, 1, 256, 128), (2, 1, 129, 64), (1, 2, 33, 5)])
def test_attention_backward_unit_scale_forms_on_the_product_layout(B, H, Nq, Nk, monkeypatch):
    """The second-form passes (`attn_bwd_dkv2_kernel`, `attn_bwd_dq2_kernel<4,2>` and `<8,4>`) that the self-attention of the DiT
    takes: q pre-scaled into log2 units (scale = ln 2), q / k as slots of a [B, N, 2, H, D] buffer and v as slot 2 of a packed
    [B, N, 3, H, D] tensor (UNEQUAL K / V row strides), gradients written into strided views, ragged last tiles, two batches;
    asm-issued LDS-DMA with hand-placed waits, so both pass-B forms must also agree bit for bit and repeat exactly."""
    import math
    from lcv_hip import ops
    D = 128
    g = torch.Generator().manual_seed(B * 1000 + Nq)
    qk = torch.zeros(B, max(Nq, Nk), 2, H, D, dtype=BF16)
    qkv = torch.zeros(B, Nk, 3, H, D, dtype=BF16)
    qs = torch.randn(B, Nq, H, D, generator=g) * 0.3
    qk[:, :Nq, 0] = (qs * (D ** -0.5 * math.log2(math.e))).to(BF16)
    qk[:, :Nk, 1] = (torch.randn(B, Nk, H, D, generator=g) * 0.3).to(BF16)
    qkv[:, :, 2] = torch.randn(B, Nk, H, D, generator=g).to(BF16)
    do = torch.randn(B, Nq, H, D, generator=g).to(BF16)
    qk_d, qkv_d, do_d = qk.to(DEV), qkv.to(DEV), do.to(DEV)
    q, k, v = qk_d[:, :Nq, 0], qk_d[:, :Nk, 1], qkv_d[:, :, 2]
    o, lse = ops.attention(q, k, v, ops.LN2, need_lse=True)
    outs = {}
    for waves in ("4", "8"):
        monkeypatch.setenv("LCV_ATTN_BWD_DQ_WAVES", waves)
        dqk = torch.full((B, max(Nq, Nk), 2, H, D), 7.0, dtype=BF16, device=DEV)       # every row must be overwritten
        dqkv = torch.full((B, Nk, 3, H, D), 7.0, dtype=BF16, device=DEV)
        dq, dk, dv = dqk[:, :Nq, 0], dqk[:, :Nk, 1], dqkv[:, :, 2]
        ops.attention_bwd(q, k, v, o, do_d, lse, dq, dk, dv, ops.LN2)
        again = [t.clone() for t in (dq, dk, dv)]
        ops.attention_bwd(q, k, v, o, do_d, lse, dq, dk, dv, ops.LN2)
        assert all(torch.equal(a, b) for a, b in zip(again, (dq, dk, dv)))
        assert torch.equal(dqkv[:, :, :2], torch.full_like(dqkv[:, :, :2], 7.0))           # neighbours of the strided views untouched
        outs[waves] = [t.clone() for t in (dq, dk, dv)]
        dk2, dv2 = dk.clone(), dv.clone()
        ops.attention_bwd(q, k, v, o, do_d, lse, dq, dk2, dv2, ops.LN2, accumulate_kv=True)
        assert rel_l2(dk2, 2 * dk.float(), bound=1e-2) < 1e-2 and rel_l2(dv2, 2 * dv.float(), bound=1e-2) < 1e-2
    assert all(torch.equal(a, b) for a, b in zip(outs["4"], outs["8"]))
    # fp32 autograd on the same bf16 inputs: softmax((q2 . k) ln 2) with q2 the pre-scaled q; dq is the gradient w.r.t. q2
    qf = qk[:, :Nq, 0].float().permute(0, 2, 1, 3).requires_grad_(True)
    kf = qk[:, :Nk, 1].float().permute(0, 2, 1, 3).requires_grad_(True)
    vf = qkv[:, :, 2].float().permute(0, 2, 1, 3).requires_grad_(True)
    ref = torch.softmax((qf @ kf.transpose(-1, -2)) * math.log(2.0), -1) @ vf
    ref.backward(do.float().permute(0, 2, 1, 3))
    dq, dk, dv = outs["4"]
    assert rel_l2(dq.permute(0, 2, 1, 3), qf.grad, bound=4.0e-3) < 4.0e-3
    assert rel_l2(dk.permute(0, 2, 1, 3), kf.grad, bound=4.2e-3) < 4.2e-3
    assert rel_l2(dv.permute(0, 2, 1, 3), vf.grad, bound=3.6e-3) < 3.6e-3
    # ... and the checker at these forms' rounding points (dS' = P (dP - delta) rounded BEFORE the scale)
    sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
    from oracle import dit_oracle as orc
    rq, rk, rv = orc.sdpa_backward_at_kernel_rounding(qk[:, :Nq, 0].permute(0, 2, 1, 3), qk[:, :Nk, 1].permute(0, 2, 1, 3),
                                                      qkv[:, :, 2].permute(0, 2, 1, 3), do.permute(0, 2, 1, 3), math.log(2.0),
                                                      scale_inside=False, o=o.cpu().permute(0, 2, 1, 3), lse=lse.cpu())
    assert rel_l2(dq.permute(0, 2, 1, 3), rq, bound=2.6e-4) < 2.6e-4
    assert rel_l2(dk.permute(0, 2, 1, 3), rk, bound=3.0e-4) < 3.0e-4
    assert rel_l2(dv.permute(0, 2, 1, 3), rv, bound=3.0e-4) < 3.0e-4


def test_norm_gate_swiglu_backward():
    from lcv_hip import ops
    from oracle import dit_oracle as orc
    import torch.nn.functional as F
    B, T, S, C = 2, 3, 9, 512
    x = _randn(B, T * S, C, seed=5); dy = _randn(B, T * S, C, seed=6)
    mod = _randn(B, T, 6 * C, seed=7, scale=0.5, dtype=torch.float32)
    dx, dmod = ops.adaln_modulate_bwd(x.to(DEV), mod.to(DEV), dy.to(DEV), 0, 1, T, need_dmod=True)
    xf = x.float().requires_grad_(True); mf = mod.clone().requires_grad_(True)
    y = orc.modulate_fp32(xf.view(B, T, S, C), mf[..., :C].unsqueeze(2), mf[..., C:2 * C].unsqueeze(2))
    y.backward(dy.float().view(B, T, S, C))
    assert rel_l2(dx, xf.grad, bound=2.5e-3) < 2.5e-3 and rel_l2(dmod, mf.grad, bound=1.0e-6) < 1.0e-6
    # affine LayerNorm with parameter grads
    w = _randn(C, seed=8); b = _randn(C, seed=9)
    dx, dw, db = ops.layernorm_affine_bwd(x.to(DEV), w.to(DEV), dy.to(DEV), need_dw=True)
    xf = x.float().requires_grad_(True); wf = w.float().requires_grad_(True); bf = b.float().requires_grad_(True)
    orc.layernorm_fp32(xf, wf, bf).backward(dy.float())
    assert rel_l2(dx, xf.grad, bound=2.5e-3) < 2.5e-3 and rel_l2(dw, wf.grad, bound=1.0e-6) < 1.0e-6 and rel_l2(db, bf.grad, bound=1.0e-6) < 1.0e-6
    # gated residual
    yv = _randn(B, T * S, C, seed=10)
    dyy, dmod = ops.gate_residual_bwd(yv.to(DEV), mod.to(DEV), dy.to(DEV), 2, T, need_dmod=True)
    g = mod[..., 2 * C:3 * C].unsqueeze(2)
    assert rel_l2(dyy, (g * dy.float().view(B, T, S, C)).view(B, T * S, C), bound=2.5e-3) < 2.5e-3
    ref_dg = (dy.float() * yv.float()).view(B, T, S, C).sum(2)
    assert rel_l2(dmod[..., 2 * C:3 * C], ref_dg, bound=1.0e-6) < 1.0e-6 and dmod[..., :2 * C].abs().max().item() == 0
    # swiglu
    gte = _randn(40, 256, seed=11); up = _randn(40, 256, seed=12); dout = _randn(40, 256, seed=13)
    dg, du = ops.swiglu_bwd(gte.to(DEV), up.to(DEV), dout.to(DEV))
    gf = gte.float().requires_grad_(True); uf = up.float().requires_grad_(True)
    (F.silu(gf) * uf).backward(dout.float())
    assert rel_l2(dg, gf.grad, bound=3.7e-3) < 3.7e-3 and rel_l2(du, uf.grad, bound=3.7e-3) < 3.7e-3


def test_fused_swiglu_training_path_matches_the_unfused_form():
    """FeedForwardSwiGLU under autograd with FROZEN weights (LoRA / delta TTA): one GEMM through the interleaved (w1, w3) copy
    whose epilogue also keeps the pre-activation rows, `lcv_swiglu_bwd_interleaved`, one GEMM against the transposed
    interleaved weight.  Against the unfused form (w1, w3, swiglu; two dx GEMMs and an add): the forward is the same bits,
    dx differs only by the rounding of the two partial dx the unfused form adds in bf16; both against fp32 autograd."""
    from lcv_hip import autograd_ops as A
    from longcat_video.modules.layers import FeedForwardSwiGLU
    M, C = 1000, 256
    ffn = FeedForwardSwiGLU(C, 4 * C, device=DEV, dtype=BF16)
    g = torch.Generator().manual_seed(5)
    for p in ffn.parameters():
        p.data.copy_((torch.randn(p.shape, generator=g) * 0.05).to(BF16))
        p.requires_grad = False
    x = _randn(M, C, seed=6).to(DEV)
    dy = _randn(M, C, seed=7, scale=0.1).to(DEV)
    xa = x.clone().requires_grad_(True)
    ya = ffn(xa)                                                   # fused (weights frozen, grad enabled)
    ya.backward(dy)
    assert ffn._w13 is not None and ffn._w13.shape == (2 * ffn.hidden_dim, C)
    xb = x.clone().requires_grad_(True)
    hb = A.swiglu(ffn.w1(xb), ffn.w3(xb))                          # the unfused form, spelled out
    yb = ffn.w2(hb)
    yb.backward(dy)
    assert torch.equal(ya, yb)
    xr = x.float().clone().requires_grad_(True)
    w1, w2, w3 = (m.weight.float() for m in (ffn.w1, ffn.w2, ffn.w3))
    gr = (xr @ w1.t()).to(BF16).float(); ur = (xr @ w3.t()).to(BF16).float()
    # straight-through rounding for the reference gradient: same values, smooth graph
    g_s = xr @ w1.t(); u_s = xr @ w3.t()
    h_ref = torch.nn.functional.silu(g_s) * u_s
    (h_ref @ w2.t()).backward(dy.float())
    e_fused, e_unfused = rel_l2(xa.grad, xr.grad), rel_l2(xb.grad, xr.grad)
    print(f"dx rel-L2 vs fp32 autograd: fused {e_fused:.2e}, unfused {e_unfused:.2e}; fused vs unfused {rel_l2(xa.grad, xb.grad):.2e}")
    assert e_fused < 1e-2 and e_fused <= 1.2 * e_unfused + 1e-4
    # trainable gate / up weights never take the fused form under autograd (it has no dW)
    ffn.w1.weight.requires_grad = True
    xc = x.clone().requires_grad_(True)
    ffn(xc).backward(dy)
    assert ffn.w1.weight.grad is not None and rel_l2(xc.grad, xb.grad, bound=1e-6) < 1e-6


def test_qknorm_rope_backward():
    from lcv_hip import ops
    from oracle import dit_oracle as orc
    grid, H, D, B = (2, 3, 4), 2, 128, 1
    N = 24
    qkv = _randn(B, N, 3, H, D, seed=14)
    wq = (1 + 0.1 * _randn(D, seed=15).float()).to(BF16); wk = (1 + 0.1 * _randn(D, seed=16).float()).to(BF16)
    dq = _randn(B, N, H, D, seed=17); dk = _randn(B, N, H, D, seed=18)
    cs = orc.rope_cos_sin_table(grid, D)
    d = qkv.to(DEV)
    dqi = torch.empty(B, N, H, D, dtype=BF16, device=DEV); dki = torch.empty_like(dqi)
    ops.qknorm_rope_bwd(d[:, :, 0], d[:, :, 1], dq.to(DEV), dk.to(DEV), dqi, dki, wq.to(DEV), wk.to(DEV), cs.to(DEV))
    ang = orc.rope_angles_3d(grid, D)
    for idx, w, dout, got in ((0, wq, dq, dqi), (1, wk, dk, dki)):
        src = qkv[:, :, idx].float().permute(0, 2, 1, 3).requires_grad_(True)
        out = orc.apply_rope(orc.rmsnorm_fp32(src, w), ang)
        out.backward(dout.float().permute(0, 2, 1, 3))
        assert rel_l2(got.permute(0, 2, 1, 3), src.grad, bound=2.5e-3) < 2.5e-3, idx
    # q_scale: the forward multiplied q by c, so dq_in scales by c; dk_in does not
    c = ops.log2_qscale(D ** -0.5)
    dqi2 = torch.empty_like(dqi); dki2 = torch.empty_like(dqi)
    ops.qknorm_rope_bwd(d[:, :, 0], d[:, :, 1], dq.to(DEV), dk.to(DEV), dqi2, dki2, wq.to(DEV), wk.to(DEV), cs.to(DEV),
                        q_scale=c)
    assert rel_l2(dqi2, dqi.float() * c, bound=3.5e-3) < 3.5e-3 and torch.equal(dki2, dki)


def test_linear_f32_backward_and_tn_skinny_and_unpatchify():
    from lcv_hip import ops
    import torch.nn.functional as F
    a = _randn(26, 512, seed=19, dtype=torch.float32); w = _randn(1536, 512, seed=20, scale=0.05)
    dy = _randn(26, 1536, seed=21, dtype=torch.float32)
    da = ops.linear_f32_smallm_bwd(dy.to(DEV), w.to(DEV), a.to(DEV), act_in=1)
    af = a.clone().requires_grad_(True)
    (F.silu(af) @ w.float().t()).backward(dy)
    assert rel_l2(da, af.grad, bound=1.0e-6) < 1.0e-6
    g = _randn(300, 64, seed=22); x = _randn(300, 4096, seed=23)
    out = ops.tn_skinny(g.to(DEV), x.to(DEV), 8, scale=2.0)
    assert rel_l2(out, 2.0 * g[:, :8].float().t() @ x.float(), bound=1.0e-6) < 1.0e-6
    out = ops.tn_skinny(g.to(DEV), x.to(DEV), 20, scale=1.0)
    assert rel_l2(out, g[:, :20].float().t() @ x.float(), bound=1.0e-6) < 1.0e-6
    dout = _randn(2, 16, 3, 8, 12, seed=24, dtype=torch.float32)
    dtok = ops.unpatchify_bwd(dout.to(DEV), 16, 3, 8, 12)
    from oracle import dit_oracle as orc
    t = torch.zeros(2, 3 * 4 * 6, 64, requires_grad=True)
    orc.unpatchify(t, 3, 4, 6, (1, 2, 2), 16).backward(dout)
    assert torch.equal(dtok.cpu(), t.grad)


def test_lora_linear_matches_reference_fixture():
    """Fused LoRALinear forward/backward against vectors produced by the reference's own LoRALinear."""
    import torch.nn as nn
    from tta.lora import LoRALinear
    from longcat_video.modules.layers import HipLinear
    t = torch.load(G / "tta_tensors.pt")["lora_linear_bf16"]
    base = HipLinear(64, 96, device=DEV, dtype=BF16)
    with torch.no_grad():
        base.weight.copy_(t["W"]); base.bias.copy_(t["b"])
    base.weight.requires_grad_(False); base.bias.requires_grad_(False)
    lora = LoRALinear(base, rank=4, alpha=16.0).to(device=DEV, dtype=BF16)
    with torch.no_grad():
        lora.lora_down.weight.copy_(t["A"]); lora.lora_up.weight.copy_(t["B"])
    x = t["x"].to(DEV).requires_grad_(True)
    y = lora(x)
    y.backward(t["gy"].to(DEV))
    assert abs(lora.scaling - float(t["scaling"])) < 1e-12
    # the reference rounds after every bf16 op; the fused path accumulates in fp32 -> compare within bf16 noise
    assert rel_l2(y, t["y"].float(), bound=4.4e-3) < 4.4e-3
    assert rel_l2(x.grad, t["dx"].float(), bound=4.3e-3) < 4.3e-3
    assert rel_l2(lora.lora_down.weight.grad, t["dA"].float(), bound=1e-2) < 1e-2
    assert rel_l2(lora.lora_up.weight.grad, t["dB"].float(), bound=1e-2) < 1e-2


def test_fused_adamw_clip_matches_reference_trace():
    from lcv_hip.ops import FusedAdamWClip
    tr = torch.load(G / "tta_tensors.pt")["adamw_trace"]
    ps = [torch.nn.Parameter(p.to(DEV)) for p in tr["init"]]
    opt = FusedAdamWClip(ps, lr=2e-3, betas=(0.9, 0.999), weight_decay=0.01, eps=1e-8)
    mism = tot = 0
    for step in range(len(tr["grads"])):
        if step < 3:
            for pg in opt.param_groups:
                pg["lr"] = 2e-3 * (step + 1) / 3
        for p, g in zip(ps, tr["grads"][step]):
            p.grad = g.to(DEV).clone()
        n = opt.clip_grad_norm_(1.0)
        opt.step()
        assert abs(n.item() - float(tr["norms"][step])) <= 2 ** -7 * float(tr["norms"][step])
        for p, exp in zip(ps, tr["after"][step]):
            d = (p.detach().cpu().view(torch.int16).int() - exp.view(torch.int16).int()).abs()
            assert d.max().item() <= 1, (step, d.max().item())   # at most one bf16 ulp apart
            mism += (d != 0).sum().item(); tot += d.numel()
    assert mism / tot < 0.02, f"{mism}/{tot} elements differ from torch's AdamW"


def _small_dit(depth=2):
    from oracle import dit_oracle as orc
    from longcat_video.modules.longcat_video_dit import LongCatVideoTransformer3DModel
    cfg = orc.small_config(hidden_size=256, depth=depth, num_heads=2, caption_channels=64)
    P = orc.make_params(cfg, seed=21, std=0.05)
    m = LongCatVideoTransformer3DModel(device=DEV, dtype=BF16, hidden_size=256, depth=depth, num_heads=2,
                                       caption_channels=64, adaln_tembed_dim=64)
    m.load_state_dict(P, strict=False)
    return m, cfg, P


@pytest.mark.parametrize("ncond,ckpt", [(1, False), (0, False), (1, True), (0, True)])
def test_dit_lora_gradients_match_oracle_autograd(ncond, ckpt):
    """loss.backward() through the HIP DiT with fused LoRA adapters vs torch autograd over the fp32 oracle with the
    same adapters folded in as W + s*B*A (mathematically identical)."""
    import functools
    from torch.utils.checkpoint import checkpoint
    from oracle import dit_oracle as orc
    from tta.lora import inject_lora_into_dit, get_lora_parameters
    from tta.flow_matching import fm_mse_loss
    m, cfg, P = _small_dit()
    for p in m.parameters():
        p.requires_grad = False
    mods = inject_lora_into_dit(m, rank=4, alpha=8.0, target_modules=["qkv", "proj"], target_ffn=True)
    g = torch.Generator().manual_seed(5)
    with torch.no_grad():
        for lm in mods:
            lm.lora_down.weight.copy_((torch.randn(lm.lora_down.weight.shape, generator=g) * 0.05).to(BF16))
            lm.lora_up.weight.copy_((torch.randn(lm.lora_up.weight.shape, generator=g) * 0.05).to(BF16))
    if ckpt:  # the reference's gradient-checkpointing switch (run_lora_tta.py:806-811)
        m.gradient_checkpointing = True
        m._gradient_checkpointing_func = functools.partial(checkpoint, use_reentrant=False)
    m.train()
    B, T, H, W, L = 1, 3, 8, 8, 16
    hs = _randn(B, 16, T, H, W, seed=30); y = _randn(B, 1, L, 64, seed=31)
    mask = torch.zeros(B, L, dtype=torch.int64); mask[:, :11] = 1
    ts = torch.zeros(B, T); ts[:, ncond:] = 431.0
    eps = _randn(B, 16, T - ncond, H, W, seed=32); x0 = _randn(B, 16, T - ncond, H, W, seed=33)
    pred = m(hs.to(DEV), ts.to(BF16).to(DEV), y.to(DEV), mask.to(DEV), num_cond_latents=ncond)
    loss = fm_mse_loss(pred, eps.to(DEV), x0.to(DEV), ncond)
    loss.backward()
    # ---- oracle with folded adapters
    names = []
    for i in range(cfg["depth"]):
        b = f"blocks.{i}."
        names += [b + "attn.qkv", b + "attn.proj", b + "cross_attn.q_linear", b + "cross_attn.kv_linear",
                  b + "cross_attn.proj", b + "ffn.w1", b + "ffn.w2", b + "ffn.w3"]
    P2 = {k: v.float() for k, v in P.items()}
    leaves = []
    for n, lm in zip(names, mods):
        A_ = lm.lora_down.weight.detach().float().cpu().requires_grad_(True)
        B_ = lm.lora_up.weight.detach().float().cpu().requires_grad_(True)
        P2[n + ".weight"] = P2[n + ".weight"] + lm.scaling * (B_ @ A_)
        leaves += [A_, B_]
    ref = orc.dit_forward(P2, cfg, hs, ts.to(BF16), y, mask, ncond, bf16=False)
    ref_loss = torch.nn.functional.mse_loss(ref[:, :, ncond:], (eps - x0).float())
    ref_loss.backward()
    assert abs(loss.item() - ref_loss.item()) < 2e-2 * abs(ref_loss.item())
    got = get_lora_parameters(mods)
    errs = [rel_l2(p.grad, l.grad) for p, l in zip(got, leaves)]
    print("max / median LoRA grad rel-L2:", max(errs), sorted(errs)[len(errs) // 2])
    print([f"{n.split('blocks.')[1]}:{e1:.3f}/{e2:.3f}" for n, e1, e2 in zip(names, errs[0::2], errs[1::2])])
    assert max(errs) < 3e-2 and sorted(errs)[len(errs) // 2] < 1.5e-2      # measured 1.3e-2 / 8.4e-3 (round 2)


def test_conditioned_loss_with_an_empty_conditioning_block():
    """`tta_total_frames = tta_context_frames = 2` splits 0 / 1 / 0 - `cond_latents` is an EMPTY [B, 16, 0, h, w]
    tensor, N_cond = 0 and the DiT runs with num_cond_latents = 0 (common.py:1388, 448-482; configs/exp3_train_frames_lora.yaml:29-33).
    The conditioned loss must accept it: same sigma / noise draws as the oracle's restatement of the reference's input builder,
    loss and LoRA gradients against oracle autograd."""
    from oracle import dit_oracle as orc
    from oracle import tta_oracle as T
    from tta.lora import inject_lora_into_dit, get_lora_parameters
    from tta.flow_matching import compute_flow_matching_loss_conditioned
    m, cfg, P = _small_dit()
    for p_ in m.parameters():
        p_.requires_grad = False
    mods = inject_lora_into_dit(m, rank=4, alpha=8.0, target_modules=["qkv", "proj"])
    g = torch.Generator().manual_seed(9)
    with torch.no_grad():
        for lm in mods:
            lm.lora_down.weight.copy_((torch.randn(lm.lora_down.weight.shape, generator=g) * 0.05).to(BF16))
            lm.lora_up.weight.copy_((torch.randn(lm.lora_up.weight.shape, generator=g) * 0.05).to(BF16))
    m.train()
    B, H, W, L = 1, 8, 8, 16
    cond = torch.zeros(B, 16, 0, H, W)
    target = _randn(B, 16, 1, H, W, seed=41, dtype=torch.float32)
    y = _randn(B, 1, L, 64, seed=42); mask = torch.zeros(B, L, dtype=torch.int64); mask[:, :9] = 1
    torch.manual_seed(123)
    loss = compute_flow_matching_loss_conditioned(m, cond.to(DEV), target.to(DEV), y.to(DEV), mask.to(DEV), device=DEV)
    loss.backward()
    # the same draws (the loss draws sigma with torch.rand and the noise with torch.randn_like on the device, in that order)
    torch.manual_seed(123)
    sigma = torch.rand(B, device=DEV, dtype=torch.float32) * (1.0 - 0.001) + 0.001
    eps = torch.randn_like(target.to(DEV))
    hs, ts, n_cond = T.build_conditioned_inputs(cond, target, sigma.cpu(), eps.cpu(), patch_t=1)
    assert n_cond == 0 and hs.shape == (B, 16, 1, H, W)
    names = T.lora_target_names(cfg["depth"], ("qkv", "proj"))      # injection order of the reference (run_lora_tta.py:286-359)
    assert len(names) == len(mods)
    P2 = {k: v.float() for k, v in P.items()}
    leaves = []
    for n, lm in zip(names, mods):
        A_ = lm.lora_down.weight.detach().float().cpu().requires_grad_(True)
        B_ = lm.lora_up.weight.detach().float().cpu().requires_grad_(True)
        P2[n + ".weight"] = P2[n + ".weight"] + lm.scaling * (B_ @ A_)
        leaves += [A_, B_]
    ref = orc.dit_forward(P2, cfg, hs, ts, y, mask, 0, bf16=False)
    ref_loss = T.conditioned_loss(ref, eps.cpu().float(), target.float(), 0)
    ref_loss.backward()
    assert abs(loss.item() - ref_loss.item()) < 2e-2 * abs(ref_loss.item()), (loss.item(), ref_loss.item())
    errs = [rel_l2(p_.grad, l.grad) for p_, l in zip(get_lora_parameters(mods), leaves)]
    assert max(errs) < 3e-2 and sorted(errs)[len(errs) // 2] < 1.5e-2, errs


def test_inner_loop_matches_reference_run():
    """finetune_lora_on_conditioning (fused LoRA + fused clip/AdamW) against the losses and final adapter weights the
    reference's own loop produced on the toy DiT of tests/golden/make_golden.py (same injected sigma / eps)."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("make_golden", G / "make_golden.py")
    mg = importlib.util.module_from_spec(spec); spec.loader.exec_module(mg)
    from tta.lora import inject_lora_into_dit, get_lora_parameters
    from tta.inner_loop import finetune_lora_on_conditioning
    t = torch.load(G / "tta_tensors.pt")["inner_loop"]
    inj = json.loads((G / "lora_injection.json").read_text())
    dit = mg.ToyDiT(dtype=BF16)
    # the fixture was saved after injection: wrapped linears appear as `<name>.original.<param>`
    dit.load_state_dict({k.replace(".original.", "."): v for k, v in t["base_state"].items()}, strict=True)
    dit.to(DEV)
    for p in dit.parameters():
        p.requires_grad = False
    hp = t["hp"]
    mods = inject_lora_into_dit(dit, rank=hp["rank"], alpha=hp["alpha"], target_modules=["qkv", "proj"],
                                target_ffn=True, target_blocks="last_2")
    order = [n for n, mm in dit.named_modules() if type(mm).__name__ == "LoRALinear"]
    assert order == inj["named_modules_order"] and len(mods) == inj["n_modules"]
    params = get_lora_parameters(mods)
    assert [list(p.shape) for p in params] == inj["param_shapes"]
    with torch.no_grad():
        for p, v in zip(params, t["init_params"]):
            p.copy_(v)
    cnt = {"i": 0}
    real_rand, real_randn_like = torch.rand, torch.randn_like
    torch.rand = lambda *a, **k: t["sig_u"][cnt["i"]].to(DEV).clone()

    def fake_randn_like(x, **k):
        e = t["eps"][cnt["i"]].to(DEV).clone(); cnt["i"] += 1
        return e
    torch.randn_like = fake_randn_like
    try:
        res = finetune_lora_on_conditioning(dit, mods, t["cond"].to(DEV), t["target"].to(DEV), None, None,
                                            num_steps=hp["num_steps"], lr=hp["lr"], warmup_steps=hp["warmup_steps"],
                                            weight_decay=hp["weight_decay"], max_grad_norm=hp["max_grad_norm"],
                                            device=DEV, dtype=BF16, early_stopper=None)
    finally:
        torch.rand, torch.randn_like = real_rand, real_randn_like
    got, exp = torch.tensor(res["losses"]), t["losses"]
    print("losses", got.tolist(), exp.tolist())
    assert torch.allclose(got, exp, rtol=3e-2, atol=1e-3)
    errs = [rel_l2(p, e.float()) for p, e in zip(params, t["final_params"])]
    print("final adapter rel-L2 (max):", max(errs))
    assert max(errs) < 0.06  # bf16 adapters after 5 Adam steps (sign-like updates amplify 1-ulp gradient noise)
    assert set(res) == {"losses", "train_time", "es_check_time", "early_stopping_info"}


def test_delta_a_gradient_and_step():
    """delta-A: d loss / d delta through the hooked t_embedder, the fp32 adaLN island and the modulation-table gradients,
    against torch autograd over the fp32 oracle; then one fused clip + AdamW(fp32, eps=1e-15) step vs torch.optim.AdamW."""
    from oracle import dit_oracle as orc
    from tta.delta import DeltaAWrapper
    from tta.flow_matching import fm_mse_loss
    from lcv_hip.ops import FusedAdamWClip
    m, cfg, P = _small_dit()
    w = DeltaAWrapper(m, adaln_tembed_dim=cfg["adaln_tembed_dim"]).to(DEV)
    with torch.no_grad():
        w.delta.copy_(0.05 * torch.randn(cfg["adaln_tembed_dim"], generator=torch.Generator().manual_seed(3)))
    d0 = w.delta.detach().clone()
    w.train()
    B, T, H, W, L, ncond = 1, 3, 8, 8, 16, 1
    hs = _randn(B, 16, T, H, W, seed=40); y = _randn(B, 1, L, 64, seed=41)
    mask = torch.zeros(B, L, dtype=torch.int64); mask[:, :11] = 1
    ts = torch.zeros(B, T); ts[:, ncond:] = 700.0
    eps = _randn(B, 16, T - ncond, H, W, seed=42); x0 = _randn(B, 16, T - ncond, H, W, seed=43)
    pred = w(hs.to(DEV), ts.to(BF16).to(DEV), y.to(DEV), mask.to(DEV), num_cond_latents=ncond)
    loss = fm_mse_loss(pred, eps.to(DEV), x0.to(DEV), ncond)
    loss.backward()
    assert not m.t_embedder._forward_hooks            # hooks removed after the call
    dl = d0.cpu().clone().requires_grad_(True)
    ref = orc.dit_forward({k: v.float() for k, v in P.items()}, cfg, hs, ts.to(BF16), y, mask, ncond, bf16=False, t_delta=dl)
    ref_loss = torch.nn.functional.mse_loss(ref[:, :, ncond:], (eps - x0).float())
    ref_loss.backward()
    e = rel_l2(w.delta.grad, dl.grad)
    print("delta-A grad rel-L2:", e)
    assert e < 3e-2
    opt = FusedAdamWClip([w.delta], lr=1e-3, eps=1e-15)
    g = w.delta.grad.detach().clone()
    opt.clip_grad_norm_(1.0); opt.step()
    pt = torch.nn.Parameter(d0.cpu().clone()); pt.grad = g.cpu().clone()
    o2 = torch.optim.AdamW([pt], lr=1e-3, betas=(0.9, 0.999), eps=1e-15)
    torch.nn.utils.clip_grad_norm_([pt], 1.0); o2.step()
    assert torch.allclose(w.delta.detach().cpu(), pt.detach(), rtol=1e-6, atol=1e-9)


def test_series25_delta_a_equals_delta_b_one_group_and_film_grads():
    """The reference's own equivalence experiment ("Series 25", sweep_experiment/configs/series_delta_a_verify_equiv.yaml;
    experimental_report.md:302-311): one delta on the `t_embedder` output (run_delta_a.py) vs one group of per-block deltas on
    every block's `t` argument (run_delta_b.py, G = 1, timestep target).  The two differ only in the final layer's modulation
    (delta-A reaches it, delta-B's block hooks do not), which is why the reference reports agreement "within sampling noise";
    with the same delta also added to the final layer's `t` argument the equivalence is exact, and that is what is asserted
    here for the prediction and for the gradient.  Also: the FiLM corrections receive exactly the modulation-table gradient
    (a zero correction leaves the forward unchanged; its gradient is the sum over blocks of d loss / d mod)."""
    from tta.delta import DeltaAWrapper, DeltaBWrapper, FiLMAdapterWrapper
    from tta.flow_matching import fm_mse_loss
    m, cfg, P = _small_dit()
    Ct = cfg["adaln_tembed_dim"]
    B, T, H, W, L, ncond = 1, 3, 8, 8, 16, 1
    hs = _randn(B, 16, T, H, W, seed=50).to(DEV); y = _randn(B, 1, L, 64, seed=51).to(DEV)
    mask = torch.zeros(B, L, dtype=torch.int64); mask[:, :11] = 1; mask = mask.to(DEV)
    ts = torch.zeros(B, T); ts[:, ncond:] = 450.0; ts = ts.to(BF16).to(DEV)
    eps = _randn(B, 16, T - ncond, H, W, seed=52).to(DEV); x0 = _randn(B, 16, T - ncond, H, W, seed=53).to(DEV)
    d = 0.05 * torch.randn(Ct, generator=torch.Generator().manual_seed(5))
    wa = DeltaAWrapper(m, adaln_tembed_dim=Ct).to(DEV)
    wb = DeltaBWrapper(m, num_groups=1, adaln_tembed_dim=Ct, hidden_size=cfg["hidden_size"], delta_target="timestep").to(DEV)
    with torch.no_grad():
        wa.delta.copy_(d); wb.deltas[0].copy_(d)
        if wb.delta_final is not None:
            wb.delta_final.copy_(d)
    assert wb.delta_final is None   # timestep target: block hooks only
    dfin = torch.nn.Parameter(d.clone().to(DEV))
    outs, grads = [], []
    for w, params in ((wa, [wa.delta]), (wb, list(wb.deltas) + [dfin])):
        w.train()
        h = None
        if w is wb:  # complete delta-B with the final layer's share of delta-A
            h = m.final_layer.register_forward_pre_hook(lambda _m, args: (args[0], args[1] + dfin.to(args[1].dtype)) + tuple(args[2:]))
        pred = w(hs, ts, y, mask, num_cond_latents=ncond)
        if h is not None:
            h.remove()
        fm_mse_loss(pred, eps, x0, ncond).backward()
        outs.append(pred.detach())
        grads.append(sum(p.grad for p in params))   # d/d(shared delta) = sum over the places it enters
    assert rel_l2(outs[1], outs[0], bound=2e-3) < 2e-3
    e = rel_l2(grads[1], grads[0])
    print("series-25 gradient rel-L2 (delta-B G=1 vs delta-A):", e)
    assert e < 3e-2
    # FiLM: zero corrections do not change the prediction; gradients are finite and non-zero in every group
    wf = FiLMAdapterWrapper(m, num_groups=2, hidden_size=cfg["hidden_size"], film_mode="full").to(DEV)
    wf.train()
    with torch.no_grad():
        base = m(hs, ts, y, mask, num_cond_latents=ncond)
    pred = wf(hs, ts, y, mask, num_cond_latents=ncond)
    assert rel_l2(pred.detach(), base, bound=2e-3) < 2e-3
    fm_mse_loss(pred, eps, x0, ncond).backward()
    for c in wf.corrections:
        assert c.grad is not None and torch.isfinite(c.grad).all() and c.grad.abs().max() > 0
    assert not any(b.adaLN_modulation._forward_hooks for b in m.blocks)


def test_qk_norm_weight_gradients():
    """Norm-weight tuning (run_norm_tune_tta.py, --norm-target qk_norm / cross_attn_norm): gradients of the q/k RMS-norm
    weights (self- and cross-attention) and of the cross-attention pre-norm affine, against torch autograd over the fp32
    oracle with the same weights as leaves."""
    from oracle import dit_oracle as orc
    from tta.delta import collect_norm_params
    from tta.flow_matching import fm_mse_loss
    m, cfg, P = _small_dit()
    params = collect_norm_params(m, "all_norm")
    assert len(params) == len(m.blocks) * 6
    for p_ in m.parameters():
        p_.requires_grad = False
    for p_ in params:
        p_.requires_grad = True
    m.train()
    B, T, H, W, L, ncond = 1, 3, 8, 8, 16, 1
    hs = _randn(B, 16, T, H, W, seed=60); y = _randn(B, 1, L, 64, seed=61)
    mask = torch.zeros(B, L, dtype=torch.int64); mask[:, :11] = 1
    ts = torch.zeros(B, T); ts[:, ncond:] = 600.0
    eps = _randn(B, 16, T - ncond, H, W, seed=62); x0 = _randn(B, 16, T - ncond, H, W, seed=63)
    pred = m(hs.to(DEV), ts.to(BF16).to(DEV), y.to(DEV), mask.to(DEV), num_cond_latents=ncond)
    fm_mse_loss(pred, eps.to(DEV), x0.to(DEV), ncond).backward()
    Pf = {k: v.float() for k, v in P.items()}
    names = []
    for i in range(len(m.blocks)):
        names += [f"blocks.{i}.pre_crs_attn_norm.weight", f"blocks.{i}.pre_crs_attn_norm.bias", f"blocks.{i}.attn.q_norm.weight",
                  f"blocks.{i}.attn.k_norm.weight", f"blocks.{i}.cross_attn.q_norm.weight", f"blocks.{i}.cross_attn.k_norm.weight"]
    for n in names:
        Pf[n] = Pf[n].clone().requires_grad_(True)
    ref = orc.dit_forward(Pf, cfg, hs, ts.to(BF16), y, mask, ncond, bf16=False)
    torch.nn.functional.mse_loss(ref[:, :, ncond:], (eps - x0).float()).backward()
    worst = 0.0
    for n, p_ in zip(names, params):
        assert p_.grad is not None, n
        e = rel_l2(p_.grad, Pf[n].grad)
        worst = max(worst, e)
        assert e < 6e-2, (n, e)
    print("norm-weight gradient rel-L2 (max over 12 tensors):", worst)


def test_joint_clip_over_bf16_and_fp32_parameters_matches_torch():
    """`--also-tune-delta`: torch runs ONE clip_grad_norm_ + ONE AdamW over bf16 norm weights and an fp32 delta
    (run_norm_tune_tta.py:230, 258-259).  Here: one fused optimizer per dtype tied by the joint coefficient."""
    from lcv_hip.ops import FusedAdamWClip
    g = torch.Generator().manual_seed(21)
    shapes = [(128,), (4096,), (128,)]
    p16 = [torch.randn(s, generator=g).to(BF16) for s in shapes]
    p32 = [torch.randn(512, generator=g) * 0.1]
    ref = [torch.nn.Parameter(p.clone()) for p in p16 + p32]
    ropt = torch.optim.AdamW(ref, lr=1e-2, betas=(0.9, 0.999), eps=1e-15)
    mine = [torch.nn.Parameter(p.clone().to(DEV)) for p in p16 + p32]
    opts = [FusedAdamWClip(mine[:3], lr=1e-2, weight_decay=0.01, eps=1e-15), FusedAdamWClip(mine[3:], lr=1e-2, weight_decay=0.01, eps=1e-15)]
    for step in range(4):
        scale = 5.0 if step % 2 == 0 else 0.01                     # clipped and unclipped steps
        grads = [(torch.randn(p.shape, generator=g) * scale).to(p.dtype) for p in ref]
        for p, q, gr in zip(ref, mine, grads):
            p.grad = gr.clone(); q.grad = gr.clone().to(DEV)
        tn = torch.nn.utils.clip_grad_norm_(ref, 1.0)
        ropt.step()
        total = FusedAdamWClip.joint_clip_grad_norm_(opts, 1.0)
        for o in opts:
            o.step()
        assert abs(total - tn.item()) <= 4e-3 * tn.item()          # bf16 per-tensor norms on both sides
        for p, q in zip(ref, mine):
            d = (q.detach().cpu().float() - p.detach().float()).abs()
            ulp = p.detach().float().abs() * (2 ** -7 if p.dtype == BF16 else 2 ** -20) + 1e-6
            assert (d <= 2 * ulp).all(), (step, d.max())


@pytest.mark.parametrize("ncond,ckpt", [(1, False), (0, True)])
def test_full_model_gradients_match_oracle_autograd(ncond, ckpt):
    """Full-model TTA (lora_experiment/scripts/run_full_tta.py:452-453: every DiT parameter trainable): loss.backward()
    through the HIP DiT must give EVERY named parameter the gradient torch autograd gives over the fp32 oracle — dense
    weights (transposes + the NT GEMM over the token axis), biases (row sums of dY^T), the fp32-island linears (adaLN
    modulation, timestep MLP), the embedders, the final layer and the norm weights."""
    import functools
    from torch.utils.checkpoint import checkpoint
    from oracle import dit_oracle as orc
    from tta.flow_matching import fm_mse_loss
    m, cfg, P = _small_dit()
    for p in m.parameters():
        p.requires_grad = True
    if ckpt:
        m.gradient_checkpointing = True
        m._gradient_checkpointing_func = functools.partial(checkpoint, use_reentrant=False)
    m.train()
    B, T, H, W, L = 1, 3, 8, 8, 16
    hs = _randn(B, 16, T, H, W, seed=40); y = _randn(B, 1, L, 64, seed=41)
    mask = torch.zeros(B, L, dtype=torch.int64); mask[:, :11] = 1
    ts = torch.zeros(B, T); ts[:, ncond:] = 612.0
    eps = _randn(B, 16, T - ncond, H, W, seed=42); x0 = _randn(B, 16, T - ncond, H, W, seed=43)
    pred = m(hs.to(DEV), ts.to(BF16).to(DEV), y.to(DEV), mask.to(DEV), num_cond_latents=ncond)
    loss = fm_mse_loss(pred, eps.to(DEV), x0.to(DEV), ncond)
    loss.backward()
    P2 = {k: v.float().clone().requires_grad_(True) for k, v in P.items()}
    ref = orc.dit_forward(P2, cfg, hs, ts.to(BF16), y, mask, ncond, bf16=False)
    ref_loss = torch.nn.functional.mse_loss(ref[:, :, ncond:], (eps - x0).float())
    ref_loss.backward()
    assert abs(loss.item() - ref_loss.item()) < 2e-2 * abs(ref_loss.item())
    errs, missing = {}, []
    for name, p in m.named_parameters():
        r = P2[name].grad
        if r is None or float(r.norm()) == 0.0:
            assert p.grad is None or float(p.grad.float().norm()) < 1e-6, name    # e.g. cross-attn k_norm with no text... never here
            continue
        if p.grad is None:
            missing.append(name)
            continue
        assert p.grad.dtype == p.dtype and p.grad.shape == p.shape, name
        errs[name] = rel_l2(p.grad, r)
    assert not missing, missing
    worst = sorted(errs.items(), key=lambda kv: -kv[1])[:6]
    print("params with a gradient:", len(errs), "worst:", [(k, round(v, 4)) for k, v in worst])
    vals = sorted(errs.values())
    assert vals[-1] < 3e-2 and vals[len(vals) // 2] < 1.5e-2, worst      # measured 1.2e-2 / 7e-3 (round 2): a wrong small term would show


def test_fused_sgd_clip_matches_torch():
    """clip_grad_norm_ + torch.optim.SGD(momentum=0, weight_decay) on bf16 tensors (run_full_tta.py:138-144, 179-180)."""
    from lcv_hip.ops import FusedSGDClip
    g = torch.Generator().manual_seed(8)
    shapes = [(300, 70), (4096,), (33,), (2049,)]
    ref = [torch.nn.Parameter((torch.randn(s, generator=g) * 0.5).to(BF16)) for s in shapes]
    mine = [torch.nn.Parameter(p.detach().clone().to(DEV)) for p in ref]
    ropt = torch.optim.SGD(ref, lr=1e-2, momentum=0.0, weight_decay=0.01)
    opt = FusedSGDClip(mine, lr=1e-2, weight_decay=0.01)
    for step in range(5):
        if step < 2:                                   # the reference's linear warm-up writes param_groups[...]["lr"]
            for o in (ropt, opt):
                for pg in o.param_groups:
                    pg["lr"] = 1e-2 * (step + 1) / 2
        scale = 4.0 if step % 2 == 0 else 0.02
        for p, q in zip(ref, mine):
            gr = (torch.randn(p.shape, generator=g) * scale).to(BF16)
            p.grad = gr.clone(); q.grad = gr.clone().to(DEV)
        before = [p.detach().float().abs().clone() for p in ref]
        tn = torch.nn.utils.clip_grad_norm_(ref, 1.0)
        ropt.step()
        n = opt.clip_grad_norm_(1.0)
        opt.step()
        assert abs(n.item() - tn.item()) <= 2 ** -7 * tn.item()
        for p, q, b in zip(ref, mine, before):
            d = (q.detach().cpu().float() - p.detach().float()).abs()
            # one bf16 ulp of the PRE-update magnitude (an update that nearly cancels leaves a tiny value whose own ulp
            # says nothing), on a small fraction of the elements (torch holds lr / wd in double, the kernel in fp32)
            assert (d <= torch.maximum(b, p.detach().float().abs()) * 2 ** -7 + 1e-30).all() and (d > 0).float().mean() < 0.02, (step, d.max())


def test_unconditioned_losses_match_reference_run():
    """compute_flow_matching_loss / _fixed (row a3) on the GPU against what the reference's own functions returned on the toy
    DiT of tests/golden/make_golden.py with the same injected sigma / noise: the inputs handed to the model bit for bit,
    the loss to the toy model's bf16 tolerance."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("make_golden", G / "make_golden.py")
    mg = importlib.util.module_from_spec(spec); spec.loader.exec_module(mg)
    from tta.flow_matching import compute_flow_matching_loss, compute_flow_matching_loss_fixed
    U = torch.load(G / "uncond_loss.pt")
    dit = mg.ToyDiT(dtype=BF16)
    dit.load_state_dict(U["toy_dit_state"], strict=True)
    dit.to(DEV).eval()
    t = U["uncond"]
    real_rand, real_randn_like = torch.rand, torch.randn_like
    torch.rand = lambda *a, **k: t["sig_u"].to(DEV).clone()
    torch.randn_like = lambda x, **k: t["eps"].to(DEV).clone()
    try:
        with torch.no_grad():
            loss = compute_flow_matching_loss(dit, t["latents"].to(DEV), None, None, device=DEV, dtype=BF16)
    finally:
        torch.rand, torch.randn_like = real_rand, real_randn_like
    seen = dit.seen[-1]
    assert torch.equal(seen["hidden_states"].cpu(), t["hidden_states"]) and torch.equal(seen["timestep"].cpu(), t["timestep"])
    assert seen["num_cond_latents"] == 0
    assert abs(loss.item() - t["loss"].item()) < 2e-2 * t["loss"].item()
    # the fixed form seeds its noise on the device it runs on (torch.Generator(device).manual_seed(42 + draw)): the draws differ
    # from the CPU fixture's by construction, so its timesteps / call count are checked against the fixture and its value
    # against the same draws evaluated with the conditioned kernel path (T_cond = 0)
    f = U["uncond_fixed"]
    n0 = len(dit.seen)
    lf = compute_flow_matching_loss_fixed(dit, f["latents"].to(DEV), None, None, f["sigmas"], noise_draws=f["noise_draws"],
                                          device=DEV, dtype=BF16)
    calls = dit.seen[n0:]
    assert len(calls) == 4 and all(torch.equal(c["timestep"].cpu(), f["timesteps"][i]) for i, c in enumerate(calls))
    assert 0.5 * f["loss"] < lf < 2.0 * f["loss"] and lf == lf
    g0 = torch.Generator(device=DEV); g0.manual_seed(42)
    n_ref = torch.randn(f["latents"].shape, generator=g0, device=DEV, dtype=BF16)
    exp_hs = ((1.0 - 0.25) * f["latents"].to(DEV).float() + 0.25 * n_ref.float()).to(BF16)
    assert torch.equal(calls[0]["hidden_states"], exp_hs)


def test_builtin_lora_module_path_forward_and_gradients_match_oracle():
    """`--use-builtin-lora` (run_lora_tta.py:104-221, row a7): standalone LoRAModule adapters behind a patched `module.forward`
    (`org + lora_up(lora_down(x)) * multiplier * alpha_scale`, block-diagonal up-projection for the fused qkv / kv linears).
    Zero-initialised adapters leave the DiT output untouched; with random adapters the prediction and every adapter gradient
    match torch autograd over the fp32 oracle with the same adapters folded into the weights."""
    from oracle import dit_oracle as orc
    from tta.lora import (get_builtin_lora_parameters, inject_builtin_lora_into_dit, reset_builtin_lora_weights,
                          unhook_builtin_lora)
    from tta.flow_matching import fm_mse_loss
    m, cfg, P = _small_dit()
    for p in m.parameters():
        p.requires_grad = False
    B, T, H, W, L = 1, 3, 8, 8, 16
    hs = _randn(B, 16, T, H, W, seed=50); y = _randn(B, 1, L, 64, seed=51)
    mask = torch.zeros(B, L, dtype=torch.int64); mask[:, :9] = 1
    ts = torch.zeros(B, T); ts[:, 1:] = 377.0
    args = (hs.to(DEV), ts.to(BF16).to(DEV), y.to(DEV), mask.to(DEV))
    with torch.no_grad():
        base = m(*args, num_cond_latents=1)
    mods = inject_builtin_lora_into_dit(m, rank=4, alpha=8.0, target_modules=["qkv", "proj"], target_ffn=False)
    assert len(mods) == 2 * 5 and all(hasattr(b.attn.qkv, "org_forward") for b in m.blocks)
    assert [type(mm.lora_up).__name__ for mm in mods[:5]] == ["BlockDiagonalLinear", "HipLinear", "HipLinear", "BlockDiagonalLinear", "HipLinear"]
    with torch.no_grad():
        assert torch.equal(m(*args, num_cond_latents=1), base)            # B = 0: no effect, bit for bit
    g = torch.Generator().manual_seed(6)
    with torch.no_grad():
        for p in get_builtin_lora_parameters(mods):
            p.copy_((torch.randn(p.shape, generator=g) * 0.05).to(BF16))
    m.train()
    eps = _randn(B, 16, T - 1, H, W, seed=52); x0 = _randn(B, 16, T - 1, H, W, seed=53)
    pred = m(*args, num_cond_latents=1)
    loss = fm_mse_loss(pred, eps.to(DEV), x0.to(DEV), 1)
    loss.backward()
    # oracle: fold every adapter into its weight (block-diagonal up-projection = blocks stacked on the output axis, each
    # reading its own `rank` columns of the down-projection output)
    names = []
    for i in range(cfg["depth"]):
        b = f"blocks.{i}."
        names += [b + "attn.qkv", b + "attn.proj", b + "cross_attn.q_linear", b + "cross_attn.kv_linear", b + "cross_attn.proj"]
    P2 = {k: v.float() for k, v in P.items()}
    leaves = []
    for n, lm in zip(names, mods):
        A_ = lm.lora_down.weight.detach().float().cpu().requires_grad_(True)
        ups = lm.lora_up.blocks if hasattr(lm.lora_up, "blocks") else [lm.lora_up]
        Bs = [u.weight.detach().float().cpu().requires_grad_(True) for u in ups]
        r = lm.lora_dim
        delta = torch.cat([Bk @ A_[k * r:(k + 1) * r] for k, Bk in enumerate(Bs)], 0)
        P2[n + ".weight"] = P2[n + ".weight"] + lm.multiplier * lm.alpha_scale * delta
        leaves += [A_] + Bs
    ref = orc.dit_forward(P2, cfg, hs, ts.to(BF16), y, mask, 1, bf16=False)
    ref_loss = torch.nn.functional.mse_loss(ref[:, :, 1:], (eps - x0).float())
    ref_loss.backward()
    assert rel_l2(pred, ref, bound=6.8e-3) < 6.8e-3 and abs(loss.item() - ref_loss.item()) < 2e-2 * ref_loss.item()
    got = get_builtin_lora_parameters(mods)
    assert len(got) == len(leaves)
    errs = [rel_l2(p.grad, l.grad) for p, l in zip(got, leaves)]
    print("builtin LoRA grad rel-L2 max / median:", max(errs), sorted(errs)[len(errs) // 2])
    assert max(errs) < 8e-2 and sorted(errs)[len(errs) // 2] < 3e-2
    reset_builtin_lora_weights(mods)
    unhook_builtin_lora(m)
    assert not any(hasattr(mm, "org_forward") for mm in m.modules())
    with torch.no_grad():
        m.eval()
        assert torch.equal(m(*args, num_cond_latents=1), base)


def test_batch_inner_loops_round_robin():
    """finetune_lora_batch / finetune_full_batch (run_lora_tta.py:558-634, run_full_tta.py:230-306; retrieval-augmented batch
    TTA trains on the eval video + neighbours, step k on video k % n, tensors moved to the device per step).  With ONE video the
    batch loop is the single-video loop: identical losses from the same seed.  With two videos the odd steps see the second
    video: its losses differ from the single-video run exactly there."""
    from tta.full_tta import finetune_full_batch, finetune_full_on_conditioning, reset_dit_weights, snapshot_base_state
    from tta.inner_loop import finetune_lora_batch, finetune_lora_on_conditioning
    from tta.lora import inject_lora_into_dit, reset_lora_weights
    m, cfg, P = _small_dit()
    for p in m.parameters():
        p.requires_grad = False
    mods = inject_lora_into_dit(m, rank=4, alpha=8.0, target_modules=["qkv", "proj"])
    g = torch.Generator().manual_seed(9)
    vids = []
    for vi in range(2):   # host-resident, like the reference's batch entries; the second video has 3x the latent scale
        lat = (torch.randn((1, 16, 4, 8, 8), generator=g) * (1.0 + 2.0 * vi)).to(BF16)
        vids.append(dict(cond_latents=lat[:, :, :2], train_latents=lat[:, :, 2:],
                         prompt_embeds=torch.randn((1, 1, 16, 64), generator=g).to(BF16), prompt_mask=torch.ones((1, 16), dtype=torch.int64)))
    kw = dict(num_steps=4, lr=1e-2, warmup_steps=2, device=DEV, dtype=BF16)

    def single():
        v = vids[0]
        return finetune_lora_on_conditioning(m, mods, v["cond_latents"].to(DEV), v["train_latents"].to(DEV), v["prompt_embeds"].to(DEV),
                                             v["prompt_mask"].to(DEV), **kw)["losses"]
    # the loss reduction adds its per-block partials with fp32 atomics: two runs agree to a few ulps, not bit for bit
    same = lambda x, y: abs(x - y) <= 1e-2 * abs(y)   # (the steps after the first amplify the ulp-level difference through bf16 weights)
    # the adapter re-initialisation draws from the global RNG too: seed AFTER the reset
    reset_lora_weights(mods); torch.manual_seed(1); a = single()
    reset_lora_weights(mods); torch.manual_seed(1)
    b = finetune_lora_batch(m, mods, vids[:1], **kw)
    assert all(same(x, y) for x, y in zip(b["losses"], a)), (a, b["losses"])
    assert b["early_stopping_info"] is None and b["es_check_time"] == 0.0
    reset_lora_weights(mods); torch.manual_seed(1)
    c = finetune_lora_batch(m, mods, vids, **kw)["losses"]
    assert same(c[0], a[0]) and not same(c[1], a[1]) and all(x == x for x in c)
    # full-model form
    from tta.lora import remove_lora_from_dit
    remove_lora_from_dit(m)
    for p in m.parameters():
        p.requires_grad = True
    base = snapshot_base_state(m)
    v = vids[0]
    torch.manual_seed(2)
    fa = finetune_full_on_conditioning(m, v["cond_latents"].to(DEV), v["train_latents"].to(DEV), v["prompt_embeds"].to(DEV),
                                       v["prompt_mask"].to(DEV), num_steps=3, lr=1e-3, warmup_steps=1, device=DEV, dtype=BF16)["losses"]
    reset_dit_weights(m, base); torch.manual_seed(2)
    fb = finetune_full_batch(m, vids[:1], num_steps=3, lr=1e-3, warmup_steps=1, device=DEV, dtype=BF16)["losses"]
    assert all(same(x, y) for x, y in zip(fb, fa)), (fa, fb)
    reset_dit_weights(m, base); torch.manual_seed(2)
    fc = finetune_full_batch(m, vids, num_steps=3, lr=1e-3, warmup_steps=1, device=DEV, dtype=BF16, optimizer_type="adamw")["losses"]
    assert same(fc[0], fa[0]) and not same(fc[1], fa[1]) and all(x == x for x in fc), (fa, fc)
