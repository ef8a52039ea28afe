"""GPU parity: every forward kernel, through the C ABI, against the CPU oracle (oracle/dit_oracle.py).

Tolerances (stated per test): kernels whose outputs are bf16 are compared with the oracle evaluated at the
same bf16 rounding points; the remaining difference is fp32 summation order, which can flip the last bf16
bit of an element, so the bar is relative-L2 <= 2e-3 (bf16 epsilon is 7.8e-3) and max |diff| <= 2 bf16 ulps
of the output scale.  Integer/index results and the fp32-exact elementwise pieces are bit-exact.
"""
import math

import pytest
import torch

from conftest import rel_l2

pytestmark = pytest.mark.gpu

BF16 = torch.bfloat16
DEV = "cuda"


def _ops():
    from lcv_hip import ops
    return ops


def _orc():
    from oracle import dit_oracle
    return dit_oracle


def _randn(*shape, seed=0, scale=1.0, dtype=BF16):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(*shape, generator=g) * scale).to(dtype)


def test_device_is_gfx950():
    from lcv_hip import lib
    lib.call("lcv_device_check")
    assert lib.load().lcv_version() >= 1


@pytest.mark.parametrize("B,T,S,C", [(2, 3, 40, 512), (1, 2, 7, 4096), (1, 1, 5, 128)])
def test_adaln_modulate(B, T, S, C):
    ops, orc = _ops(), _orc()
    x = _randn(B, T * S, C, seed=1)
    mod = _randn(B, T, 6 * C, seed=2, scale=0.5, dtype=torch.float32)
    y = ops.adaln_modulate(x.to(DEV), mod.to(DEV), 3, 4, T)
    sh, sc = mod[..., 3 * C:4 * C].unsqueeze(2), mod[..., 4 * C:5 * C].unsqueeze(2)
    ref = orc.modulate_fp32(x.view(B, T, S, C), sh, sc, rnd=orc.bf16_round).view(B, T * S, C)
    assert rel_l2(y, ref, bound=3.1e-5) < 3.1e-5
    assert (y.float().cpu() - ref).abs().max() <= 2 * 2 ** -7 * ref.abs().max()


def test_layernorm_affine():
    ops, orc = _ops(), _orc()
    x = _randn(3, 50, 1024, seed=3)
    w = _randn(1024, seed=4); b = _randn(1024, seed=5)
    y = ops.layernorm_affine(x.to(DEV), w.to(DEV), b.to(DEV))
    ref = orc.bf16_round(orc.layernorm_fp32(x, w, b))
    assert rel_l2(y, ref, bound=1.0e-6) < 1.0e-6


def test_gate_residual():
    ops, orc = _ops(), _orc()
    B, T, S, C = 2, 3, 11, 256
    x = _randn(B, T * S, C, seed=6); y = _randn(B, T * S, C, seed=7)
    mod = _randn(B, T, 6 * C, seed=8, dtype=torch.float32)
    out = ops.gate_residual(x.to(DEV), y.to(DEV), mod.to(DEV), 2, T)
    g = mod[..., 2 * C:3 * C].unsqueeze(2)
    ref = orc.bf16_round(x.float() + (g * y.float().view(B, T, S, C)).view(B, T * S, C))
    # same fp32 expression evaluated once: only fma contraction can differ
    assert rel_l2(out, ref, bound=8.1e-6) < 8.1e-6
    out2 = ops.gate_residual(x.to(DEV), y.to(DEV), None, 0, 1)
    assert torch.equal(out2.cpu(), (x.float() + y.float()).to(BF16))


@pytest.mark.parametrize("grid,H", [((2, 3, 5), 16), ((1, 4, 4), 32)])
def test_qknorm_rope(grid, H):
    ops, orc = _ops(), _orc()
    N = grid[0] * grid[1] * grid[2]
    B, D = 2, 128
    qkv = _randn(B, N, 3, H, D, seed=9)
    wq = (1 + 0.1 * _randn(D, seed=10).float()).to(BF16); wk = (1 + 0.1 * _randn(D, seed=11).float()).to(BF16)
    cs = orc.rope_cos_sin_table(grid, D)
    d = qkv.to(DEV)
    ops.qknorm_rope(d[:, :, 0], d[:, :, 1], None, d[:, :, 0], d[:, :, 1], None, wq.to(DEV), wk.to(DEV), cs.to(DEV))
    ang = orc.rope_angles_3d(grid, D)
    for idx, w in ((0, wq), (1, wk)):
        src = qkv[:, :, idx].permute(0, 2, 1, 3)  # [B,H,N,D]
        ref = orc.apply_rope(orc.rmsnorm_fp32(src, w, rnd=orc.bf16_round), ang, orc.bf16_round)
        got = d[:, :, idx].permute(0, 2, 1, 3)
        assert rel_l2(got, ref, bound=5.6e-5) < 5.6e-5, idx
    assert torch.equal(d[:, :, 2].cpu(), qkv[:, :, 2])  # V untouched
    # q_scale: the attention scale folded into q before ITS single bf16 rounding; k unaffected
    c = ops.log2_qscale(D ** -0.5)
    d2 = qkv.to(DEV)
    ops.qknorm_rope(d2[:, :, 0], d2[:, :, 1], None, d2[:, :, 0], d2[:, :, 1], None, wq.to(DEV), wk.to(DEV), cs.to(DEV),
                    q_scale=c)
    src = qkv[:, :, 0].permute(0, 2, 1, 3)
    ref = orc.bf16_round(orc.apply_rope(orc.rmsnorm_fp32(src, wq, rnd=orc.bf16_round), ang) * c)
    assert rel_l2(d2[:, :, 0].permute(0, 2, 1, 3), ref, bound=2.9e-5) < 2.9e-5
    assert torch.equal(d2[:, :, 1].cpu(), d[:, :, 1].cpu())


@pytest.mark.parametrize("B,H,Nq,Nk", [(1, 2, 300, 300), (2, 1, 64, 77), (1, 1, 513, 1000), (1, 2, 256, 64), (1, 1, 31, 5)])
def test_attention(B, H, Nq, Nk):
    ops, orc = _ops(), _orc()
    D = 128
    q = _randn(B, Nq, H, D, seed=12); k = _randn(B, Nk, H, D, seed=13); v = _randn(B, Nk, H, D, seed=14)
    scale = D ** -0.5
    o, lse = ops.attention(q.to(DEV), k.to(DEV), v.to(DEV), scale, need_lse=True)
    ref = orc.sdpa(q.permute(0, 2, 1, 3), k.permute(0, 2, 1, 3), v.permute(0, 2, 1, 3), scale)  # fp32
    # tolerance: P is rounded to bf16 before PV (as flash-attn does) and O is stored in bf16
    assert rel_l2(o.permute(0, 2, 1, 3), ref, bound=3.6e-3) < 3.6e-3
    # ... and against the tile-by-tile restatement of the kernels' own arithmetic (deferred running max per 32-row group, P rounded
    # to bf16 against it): what is left is fp32 summation order and the bf16 flips that follow from it
    ref_r = orc.sdpa_at_kernel_rounding(q.permute(0, 2, 1, 3), k.permute(0, 2, 1, 3), v.permute(0, 2, 1, 3), scale)
    assert rel_l2(o.permute(0, 2, 1, 3), ref_r, bound=1.1e-4) < 1.1e-4
    s = (q.permute(0, 2, 1, 3).float() @ k.permute(0, 2, 1, 3).float().transpose(-1, -2)) * scale
    assert torch.allclose(lse.cpu(), torch.logsumexp(s, dim=-1), atol=2e-4, rtol=1e-5)


@pytest.mark.parametrize("B,H,Nq,Nk", [(1, 2, 300, 300), (2, 1, 64, 77), (1, 1, 513, 1000), (1, 1, 31, 5), (1, 2, 700, 1664)])
def test_attention_log2_prescaled_q(B, H, Nq, Nk):
    """scale = ln 2 with q already multiplied by head_dim^-0.5 * log2(e): the multiply-free softmax body of the long-key
    kernel (Nk > 512) and the general body (Nk <= 512) must both equal softmax(q'.k ln 2) v; one key row is spiked so the
    running max jumps by more than the deferred-rescale threshold mid-stream, and 1000 / 77 / 5 are ragged last tiles."""
    ops, orc = _ops(), _orc()
    D = 128
    c = ops.log2_qscale(D ** -0.5)
    q = (_randn(B, Nq, H, D, seed=12).float() * c).to(BF16); k = _randn(B, Nk, H, D, seed=13); v = _randn(B, Nk, H, D, seed=14)
    k[0, (2 * Nk) // 3] *= 6.0
    o, lse = ops.attention(q.to(DEV), k.to(DEV), v.to(DEV), ops.LN2, need_lse=True)
    ref = orc.sdpa(q.permute(0, 2, 1, 3), k.permute(0, 2, 1, 3), v.permute(0, 2, 1, 3), ops.LN2)
    assert rel_l2(o.permute(0, 2, 1, 3), ref, bound=2.6e-3) < 2.6e-3
    ref_r = orc.sdpa_at_kernel_rounding(q.permute(0, 2, 1, 3), k.permute(0, 2, 1, 3), v.permute(0, 2, 1, 3), ops.LN2)
    assert rel_l2(o.permute(0, 2, 1, 3), ref_r, bound=6.3e-5) < 6.3e-5
    s = (q.permute(0, 2, 1, 3).float() @ k.permute(0, 2, 1, 3).float().transpose(-1, -2)) * ops.LN2
    assert torch.allclose(lse.cpu(), torch.logsumexp(s, dim=-1), atol=2e-4, rtol=1e-5)


@pytest.mark.parametrize("B,H,Nq,Nk", [(1, 2, 600, 700), (2, 3, 257, 577), (1, 1, 64, 1024), (1, 2, 333, 641), (2, 1, 96, 3000),
                                       (1, 1, 130, 513), (1, 2, 65, 576), (1, 1, 256, 832)])   # 9 (ragged / full) and 13 key tiles
def test_attention_64_rows_per_wave_kernel_is_the_default_and_equals_the_two_wave_kernel(B, H, Nq, Nk, monkeypatch):
    """`attn_fwd_w64_kernel` (one wave per SIMD, 64 query rows per wave, asm-issued MFMAs: csrc/attn_fwd_w64.hip) takes every
    unit-scale call with more than 512 keys.  Inputs as the DiT hands them over: q / k / v as strided slots of packed buffers,
    q pre-scaled; two key rows spiked at different places so that EACH of a wave's two query blocks goes through the deferred
    rescale (the one path that moves O between AGPRs and VGPRs around asm MFMAs), ragged last tiles in keys and queries.
    Checked against fp32 softmax and against the two-waves-per-SIMD kernel (same operand maps; only the order of the fp32 row
    sum differs), and twice for the same bits."""
    ops, orc = _ops(), _orc()
    from lcv_hip import lib as L
    D = 128
    c = ops.log2_qscale(D ** -0.5)
    qk = torch.zeros(B, max(Nq, Nk), 2, H, D, dtype=BF16)
    qk[:, :Nq, 0] = (_randn(B, Nq, H, D, seed=31).float() * c).to(BF16)
    qk[:, :Nk, 1] = _randn(B, Nk, H, D, seed=32)
    qkv = torch.zeros(B, Nk, 3, H, D, dtype=BF16)
    qkv[:, :, 2] = _randn(B, Nk, H, D, seed=33)
    qk[0, Nk // 3, 1] *= 7.0                      # every query's maximum jumps here ...
    qk[0, (3 * Nk) // 4, 1, :, :] = qk[0, 5, 0, :, :].float().mul(40.0).to(BF16)   # ... again for query 5 only (block 0 of wave 0) ...
    qk[0, Nk // 2, 1, :, :] = qk[0, 40, 0, :, :].float().mul(40.0).to(BF16)        # ... and for query 40 only (block 1 of wave 0)
    qd, kd, vd = qk.to(DEV)[:, :Nq, 0], qk.to(DEV)[:, :Nk, 1], qkv.to(DEV)[:, :, 2]
    monkeypatch.delenv("LCV_ATTN_FWD_W64", raising=False)
    o, lse = ops.attention(qd, kd, vd, ops.LN2, need_lse=True)
    assert L.load().lcv_attn_fwd_last_kernel().decode() == "attn_fwd_w64_kernel"
    o_again, lse_again = ops.attention(qd, kd, vd, ops.LN2, need_lse=True)
    assert torch.equal(o, o_again) and torch.equal(lse, lse_again)
    monkeypatch.setenv("LCV_ATTN_FWD_W64", "0")
    o2, lse2 = ops.attention(qd, kd, vd, ops.LN2, need_lse=True)
    assert L.load().lcv_attn_fwd_last_kernel().decode() == "attn_fwd_pipe_kernel"
    assert rel_l2(o, o2.float(), bound=7.4e-5) < 7.4e-5 and torch.allclose(lse, lse2, atol=2e-5, rtol=1e-6)
    qf, kf, vf = qk[:, :Nq, 0].permute(0, 2, 1, 3), qk[:, :Nk, 1].permute(0, 2, 1, 3), qkv[:, :, 2].permute(0, 2, 1, 3)
    ref = orc.sdpa(qf, kf, vf, ops.LN2)
    assert rel_l2(o.permute(0, 2, 1, 3), ref, bound=2.5e-3) < 2.5e-3
    # the kernel's own arithmetic restated tile by tile (deferred running max per 32-row block, P rounded to bf16 against it): the
    # spikes above take the rescale path in the restatement exactly where the kernel takes it
    ref_r = orc.sdpa_at_kernel_rounding(qf, kf, vf, ops.LN2)
    assert rel_l2(o.permute(0, 2, 1, 3), ref_r, bound=8.9e-5) < 8.9e-5
    s = (qf.float() @ kf.float().transpose(-1, -2)) * ops.LN2
    assert torch.allclose(lse.cpu(), torch.logsumexp(s, dim=-1), atol=3e-4, rtol=1e-5)


def test_attention_fuzz_against_the_restatement_of_the_kernels_arithmetic():
    """24 seeded random (batch, heads, queries, keys) through whichever forward kernel the launcher picks (general, two-wave, 64 rows
    per wave; plain scale and q pre-scaled into log2 units), one key row spiked so the deferred rescale fires mid-stream, against
    `sdpa_at_kernel_rounding`: only fp32 summation order separates them, so the bound is 20 x tighter than against fp32 softmax."""
    ops, orc = _ops(), _orc()
    D = 128
    g = torch.Generator().manual_seed(77)
    ri = lambda lo, hi: int(torch.randint(lo, hi + 1, (1,), generator=g))
    worst = 0.0
    for case in range(24):
        B, H = ri(1, 2), ri(1, 3)
        Nq = [ri(1, 700), 32 * ri(1, 12) + ri(-1, 1), 256 * ri(1, 2) + ri(-1, 1)][case % 3]
        Nk = [ri(1, 1700), 64 * ri(1, 20) + ri(-1, 1), ri(513, 1400)][(case // 3) % 3]
        unit = case % 2 == 1
        q = _randn(B, Nq, H, D, seed=1000 + case); k = _randn(B, Nk, H, D, seed=2000 + case); v = _randn(B, Nk, H, D, seed=3000 + case)
        k[0, (2 * Nk) // 3] *= 5.0
        scale = D ** -0.5
        if unit:
            q = (q.float() * ops.log2_qscale(scale)).to(BF16)
            scale = ops.LN2
        o, _ = ops.attention(q.to(DEV), k.to(DEV), v.to(DEV), scale, need_lse=True)
        ref = orc.sdpa_at_kernel_rounding(q.permute(0, 2, 1, 3), k.permute(0, 2, 1, 3), v.permute(0, 2, 1, 3), scale)
        e = rel_l2(o.permute(0, 2, 1, 3), ref)
        worst = max(worst, e)
        assert e < 2.0e-4, (case, B, H, Nq, Nk, unit, e)
    print(f"attention fuzz: worst rel-L2 against the restatement {worst:.2e}")


def test_lse_merge_of_key_ranges_equals_the_one_call_attention():
    """Sequence parallelism's overlap path (`LCV_SP_OVERLAP=1`, lcv_hip/autograd_ops.py::_sp_attention_overlapped) attends the
    local keys while the gather is in flight and merges the partial results through their log-sum-exps.  The merge assumes
    natural-log units for the kernel's lse; this pins that on ONE GPU: K / V cut into three ranges (ragged sizes, one of them
    short enough for the short-key body), both scale conventions, merged == the one-call result up to one more bf16 rounding
    of a partial output."""
    from lcv_hip import autograd_ops as A
    ops, orc = _ops(), _orc()
    B, H, Nq, Nk, D = 2, 2, 333, 1500, 128
    for scale, c in ((D ** -0.5, 1.0), (ops.LN2, ops.log2_qscale(D ** -0.5))):
        q = (_randn(B, Nq, H, D, seed=21).float() * c).to(BF16).to(DEV)
        k = _randn(B, Nk, H, D, seed=22).to(DEV); v = _randn(B, Nk, H, D, seed=23).to(DEV)
        whole, lse = ops.attention(q, k, v, scale, need_lse=True)
        cuts = [(0, 700), (700, 777), (777, Nk)]
        parts = [ops.attention(q, k[:, a:b], v[:, a:b], scale, need_lse=True) for a, b in cuts]
        merged = A.merge_attention_parts(parts)
        lse_m = torch.logsumexp(torch.stack([l for _, l in parts], 0), dim=0)
        assert torch.allclose(lse_m, lse, atol=2e-4, rtol=1e-5)
        e = rel_l2(merged, whole.float())
        ref = orc.sdpa(q.cpu().permute(0, 2, 1, 3), k.cpu().permute(0, 2, 1, 3), v.cpu().permute(0, 2, 1, 3), scale)
        assert e < 4e-3 and rel_l2(merged.cpu().permute(0, 2, 1, 3), ref, bound=4.4e-3) < 4.4e-3, (scale, e)
        # a wrong unit (log2 lse read as natural log) would weight the parts by 2^(.) instead of e^(.): far outside this bound
        bad = A.merge_attention_parts([(o_i, l_i * 1.4426950408889634) for o_i, l_i in parts])
        assert rel_l2(bad, whole.float()) > 5 * e


def test_attention_strided_packed_qkv_and_spike():
    """q/k/v as views of a packed [B,N,3,H,D] buffer; one key row spiked so the running max jumps mid-stream."""
    ops, orc = _ops(), _orc()
    B, N, H, D = 1, 400, 2, 128
    qkv = _randn(B, N, 3, H, D, seed=15)
    qkv[0, 300, 1] *= 8.0
    d = qkv.to(DEV)
    o, _ = ops.attention(d[:, :, 0], d[:, :, 1], d[:, :, 2], D ** -0.5)
    ref = orc.sdpa(qkv[:, :, 0].permute(0, 2, 1, 3), qkv[:, :, 1].permute(0, 2, 1, 3), qkv[:, :, 2].permute(0, 2, 1, 3), D ** -0.5)
    assert rel_l2(o.permute(0, 2, 1, 3), ref, bound=2.3e-3) < 2.3e-3


@pytest.mark.parametrize("M,N,K,tile", [(300, 192, 128, "1"), (70, 64, 64, "1"), (2050, 1024, 256, "2"), (513, 768, 192, "2"), (700, 300, 1088, "2"), (2050, 1024, 256, "6"), (513, 768, 192, "6"), (700, 300, 1088, "6"), (300, 192, 128, "7"), (70, 64, 64, "7"),
                                          (2050, 1024, 256, "8"), (513, 768, 192, "8"), (700, 300, 1088, "8"), (300, 192, 128, "8"),
                                          (2050, 1024, 256, "9"), (513, 768, 192, "9"), (300, 192, 128, "9"),
                                          (2050, 1024, 256, "k"), (513, 768, 192, "k"), (300, 192, 128, "k")])
def test_gemm_nt_bias(M, N, K, tile, monkeypatch):
    monkeypatch.setenv("LCV_GEMM_TILE", tile)
    ops, orc = _ops(), _orc()
    a = _randn(M, K, seed=16); w = _randn(N, K, seed=17, scale=0.05); b = _randn(N, seed=18)
    c = ops.gemm_nt(a.to(DEV), w.to(DEV), b.to(DEV))
    ref = orc.linear(a, w, b, orc.bf16_round)
    assert rel_l2(c, ref, bound=3.6e-5) < 3.6e-5
    c32 = ops.gemm_nt(a.to(DEV), w.to(DEV), b.to(DEV), out_f32=True)
    assert rel_l2(c32, orc.linear(a, w, b), bound=1.0e-6) < 1.0e-6


@pytest.mark.parametrize("tile", ["1", "7", "8", "9", "k"])
def test_gemm_nt_lora_and_epilogues(tile, monkeypatch):
    monkeypatch.setenv("LCV_GEMM_TILE", tile)
    ops, orc = _ops(), _orc()
    from lcv_hip.lib import LCV_EPI_GATE_RESIDUAL, LCV_EPI_GELU_TANH, LCV_EPI_SWIGLU
    import torch.nn.functional as F
    M, N, K, R = 330, 256, 256, 8
    a = _randn(M, K, seed=19); w = _randn(N, K, seed=20, scale=0.05); b = _randn(N, seed=21)
    A = _randn(R, K, seed=22, scale=0.05); Bu = _randn(N, R, seed=23, scale=0.05)
    s = 2.0
    h = ops.lora_down(a.to(DEV), A.to(DEV), s)  # [M, 64]
    href = s * orc.bf16_round(a.float() @ A.float().t())
    assert rel_l2(h[:, :R], href, bound=2e-3) < 2e-3 and h[:, R:].abs().max().item() == 0
    w2 = torch.zeros(N, 64, dtype=BF16); w2[:, :R] = Bu
    c = ops.gemm_nt(a.to(DEV), w.to(DEV), b.to(DEV), a2=h, w2=w2.to(DEV))
    ref = orc.bf16_round(a.float() @ w.float().t() + b.float() + h[:, :R].float().cpu() @ Bu.float().t())
    assert rel_l2(c, ref, bound=3.2e-5) < 3.2e-5
    # gelu-tanh epilogue
    c = ops.gemm_nt(a.to(DEV), w.to(DEV), b.to(DEV), epilogue=LCV_EPI_GELU_TANH)
    ref = orc.bf16_round(F.gelu(orc.linear(a, w, b, orc.bf16_round), approximate="tanh"))
    assert rel_l2(c, ref, bound=1.2e-5) < 1.2e-5
    # gate-residual epilogue (rows_per_frame = 110 -> 3 frames)
    resid = _randn(M, N, seed=24); mod = _randn(1, 3, 6 * N, seed=25, dtype=torch.float32)
    c = ops.gemm_nt(a.to(DEV), w.to(DEV), b.to(DEV), epilogue=LCV_EPI_GATE_RESIDUAL, resid=resid.to(DEV),
                    mod=mod.to(DEV), gate_idx=5, rows_per_frame=110)
    g = mod[0, :, 5 * N:6 * N].repeat_interleave(110, dim=0)
    ref = orc.bf16_round(resid.float() + g * orc.linear(a, w, b, orc.bf16_round))
    assert rel_l2(c, ref, bound=9.4e-6) < 9.4e-6
    # SwiGLU epilogue with [32 gate | 32 up] interleaved weight rows
    Fh = 128
    w1 = _randn(Fh, K, seed=26, scale=0.05); w3 = _randn(Fh, K, seed=27, scale=0.05)
    wi = torch.stack([w1.view(Fh // 32, 32, K), w3.view(Fh // 32, 32, K)], dim=1).reshape(2 * Fh, K).contiguous()
    c = ops.gemm_nt(a.to(DEV), wi.to(DEV), None, epilogue=LCV_EPI_SWIGLU)
    gte, up = orc.linear(a, w1, None, orc.bf16_round), orc.linear(a, w3, None, orc.bf16_round)
    ref = orc.bf16_round(orc.bf16_round(F.silu(gte)) * up)
    assert c.shape == (M, Fh) and rel_l2(c, ref, bound=1.1e-5) < 1.1e-5


def test_gemm_8phase_matches_plain_schedule_bitwise(monkeypatch):
    """The 8-phase ping-pong kernel (counted vmcnt, raw barriers, staggered wave rows) accumulates K in the same order on
    the same MFMA as the one-barrier-per-tile kernel: any difference at all is a synchronisation bug (a fragment read
    before its LDS-DMA landed, or a slot restaged under a reader).  Long K, many tiles, repeated launches."""
    ops = _ops()
    M, N, K = 4096 + 70, 2048 + 30, 4096
    a = _randn(M, K, seed=61).to(DEV); w = _randn(N, K, seed=62, scale=0.05).to(DEV); b = _randn(N, seed=63).to(DEV)
    monkeypatch.setenv("LCV_GEMM_TILE", "6")
    ref = ops.gemm_nt(a, w, b)
    for tile in ("8", "9"):  # 9 = persistent workgroups (here 17 x 9 = 153 tiles on <= 256 CUs, see the next test for > 256)
        monkeypatch.setenv("LCV_GEMM_TILE", tile)
        for _ in range(5):
            assert torch.equal(ops.gemm_nt(a, w, b), ref)
    # with the rank-r pair as the last K tile
    a2 = _randn(M, 64, seed=64).to(DEV); w2 = _randn(N, 64, seed=65, scale=0.05).to(DEV)
    monkeypatch.setenv("LCV_GEMM_TILE", "6")
    ref = ops.gemm_nt(a, w, b, a2=a2, w2=w2)
    for tile in ("8", "9"):
        monkeypatch.setenv("LCV_GEMM_TILE", tile)
        for _ in range(3):
            assert torch.equal(ops.gemm_nt(a, w, b, a2=a2, w2=w2), ref)


def test_gemm_persistent_many_tiles_bitwise(monkeypatch):
    """More tiles than CUs, so every persistent workgroup walks several tiles and each tile's LDS-DMA prologue is issued
    under the previous tile's epilogue; ragged M and N edges; must equal the one-barrier kernel bit for bit."""
    ops = _ops()
    M, N, K = 256 * 37 + 19, 256 * 11 + 40, 512
    a = _randn(M, K, seed=71).to(DEV); w = _randn(N, K, seed=72, scale=0.05).to(DEV); b = _randn(N, seed=73).to(DEV)
    monkeypatch.setenv("LCV_GEMM_TILE", "6")
    ref = ops.gemm_nt(a, w, b)
    monkeypatch.setenv("LCV_GEMM_TILE", "9")
    for _ in range(4):
        assert torch.equal(ops.gemm_nt(a, w, b), ref)


@pytest.mark.parametrize("M,N,K,K2,kind", [(300, 512, 256, 0, "plain"), (256 * 3 + 19, 512, 512, 0, "plain"), (2500, 512, 192, 128, "plain"),
                                           (256 * 37 + 19, 256 * 11, 512, 0, "plain"), (4096 + 70, 2048, 4096, 64, "plain"),
                                           (4096, 1024, 1024, 0, "gate"), (4096 + 33, 1024, 1024, 64, "gate"), (4096 + 33, 1024, 1024, 0, "gate_nomod"),
                                           (4096, 2048, 1024, 0, "swiglu"), (4096 + 7, 2048, 1024, 0, "swiglu_train"), (12480, 4096, 4096, 0, "gate"),
                                           (1000, 768, 256, 0, "nobias"), (4680, 1024, 512, 0, "gate_exact_table")])
def test_gemm4k_default_kernel_bitwise_every_epilogue(M, N, K, K2, kind, monkeypatch):
    """csrc/gemm4k.h (round 4, the default for the DiT's big GEMMs): four waves x 128 x 128, 64-deep K tiles in 128-byte rows, three
    barriers per K tile, ONE epilogue with 16-byte stores on a PAIRED column layout (a permutation of the weight rows inside the
    LDS-DMA).  Same MFMA and K order as the one-barrier kernel (LCV_GEMM_TILE=6): every output bit-identical - rows past M
    predicated off (ragged M), more tiles than CUs, the rank-r K tiles, the gate row changing INSIDE a wave tile (rows_per_frame not
    a multiple of 128), no gate, no bias, SwiGLU with and without the pre-activation rows, repeated launches."""
    ops = _ops()
    from lcv_hip.lib import LCV_EPI_GATE_RESIDUAL, LCV_EPI_SWIGLU
    a = _randn(M, K, seed=1).to(DEV); w = _randn(N, K, seed=2, scale=0.05).to(DEV)
    b = None if kind == "nobias" else _randn(N, seed=3).to(DEV)
    kw = {}
    if K2:
        kw.update(a2=_randn(M, K2, seed=4).to(DEV), w2=_randn(N, K2, seed=5, scale=0.05).to(DEV))
    if kind == "gate":
        T = 4
        kw.update(epilogue=LCV_EPI_GATE_RESIDUAL, resid=_randn(M, N, seed=6).to(DEV), mod=_randn(1, T, 3 * N, seed=7, dtype=torch.float32).to(DEV),
                  gate_idx=2, rows_per_frame=(M + T - 1) // T)
    if kind == "gate_nomod":
        kw.update(epilogue=LCV_EPI_GATE_RESIDUAL, resid=_randn(M, N, seed=6).to(DEV))
    if kind == "gate_exact_table":
        # 3 frames of 1 560 rows and a table of exactly 3 gate rows that ENDS its allocation: the last 256-row tile has a wave tile
        # wholly past M (rows 4 736 ...), whose frame index must not be looked up (round 4: a GPU fault in the first form)
        mod = _randn(3 * 3 * N, seed=7, dtype=torch.float32).to(DEV).view(1, 3, 3 * N)
        kw.update(epilogue=LCV_EPI_GATE_RESIDUAL, resid=_randn(M, N, seed=6).to(DEV), mod=mod, gate_idx=2, rows_per_frame=1560)
    if kind.startswith("swiglu"):
        kw.update(epilogue=LCV_EPI_SWIGLU)
    aux_ref = None
    monkeypatch.setenv("LCV_GEMM_TILE", "6")
    if kind == "swiglu_train":
        kw.update(swiglu_aux=torch.empty(M, N, device=DEV, dtype=BF16))
    ref = ops.gemm_nt(a, w, b, **kw)
    if kind == "swiglu_train":
        aux_ref = kw["swiglu_aux"].clone()
    for tile in ("k", None):          # forced, then the dispatcher's own choice
        if tile is None:
            monkeypatch.delenv("LCV_GEMM_TILE")
        else:
            monkeypatch.setenv("LCV_GEMM_TILE", tile)
        for _ in range(2):
            if aux_ref is not None:
                kw["swiglu_aux"].zero_()
            assert torch.equal(ops.gemm_nt(a, w, b, **kw), ref), (M, N, K, K2, kind, tile)
            if aux_ref is not None:
                assert torch.equal(kw["swiglu_aux"], aux_ref)


def test_gemm_dispatch_fuzz_bitwise_against_the_one_barrier_kernel(monkeypatch):
    """40 seeded random calls through the dispatcher (which picks gemm4k wherever its epilogue serves the call and the round-3 kernels
    elsewhere) against the one-barrier kernel, bit for bit: row counts from 1 to 3 000 incl. +-1 around tile multiples, every N
    multiple of 256 up to 1 280 plus one that is not, K from 128, with / without the rank-r K tiles, bias, the gate-residual epilogue
    with frame lengths above and below the 128 rows gemm4k needs, SwiGLU.  (The one GPU fault of round 4 was an edge of exactly this
    kind: a wave tile wholly past M looking up its frame's gate row.)"""
    ops = _ops()
    from lcv_hip.lib import LCV_EPI_GATE_RESIDUAL, LCV_EPI_SWIGLU
    g = torch.Generator().manual_seed(2024)
    ri = lambda lo, hi: int(torch.randint(lo, hi + 1, (1,), generator=g))
    pick = lambda xs: xs[ri(0, len(xs) - 1)]
    for case in range(40):
        M = pick([ri(1, 3000), 256 * ri(1, 9) + pick([-1, 0, 1]), 128 * ri(1, 5) + pick([-1, 1])])
        N = pick([256, 512, 768, 1024, 1280, 384])
        K = pick([128, 192, 256, 320, 512, 1024])
        K2 = pick([0, 0, 64, 128])
        kind = pick(["plain", "nobias", "gate", "gate_short_frames", "gate_nomod", "swiglu"])
        a = _randn(M, K, seed=100 + case).to(DEV); w = _randn(N, K, seed=200 + case, scale=0.05).to(DEV)
        b = None if kind == "nobias" else _randn(N, seed=300 + case).to(DEV)
        kw = {}
        if K2:
            kw.update(a2=_randn(M, K2, seed=400 + case).to(DEV), w2=_randn(N, K2, seed=500 + case, scale=0.05).to(DEV))
        if kind in ("gate", "gate_short_frames"):
            rpf = max(1, (M + 2) // 3) if kind == "gate" else max(1, min(M, 96))
            T = (M + rpf - 1) // rpf
            kw.update(epilogue=LCV_EPI_GATE_RESIDUAL, resid=_randn(M, N, seed=600 + case).to(DEV),
                      mod=_randn(1, T, 3 * N, seed=700 + case, dtype=torch.float32).to(DEV), gate_idx=ri(0, 2), rows_per_frame=rpf)
        if kind == "gate_nomod":
            kw.update(epilogue=LCV_EPI_GATE_RESIDUAL, resid=_randn(M, N, seed=600 + case).to(DEV))
        if kind == "swiglu":
            if N % 64:
                continue
            kw.update(epilogue=LCV_EPI_SWIGLU)
        monkeypatch.setenv("LCV_GEMM_TILE", "6")
        ref = ops.gemm_nt(a, w, b, **kw)
        monkeypatch.delenv("LCV_GEMM_TILE")
        got = ops.gemm_nt(a, w, b, **kw)
        assert torch.equal(got, ref), (case, M, N, K, K2, kind)
        assert torch.equal(ops.gemm_nt(a, w, b, **kw), ref), (case, "repeat")


@pytest.mark.parametrize("M,N,K,kind", [(12480, 4096, 4096, "plain"), (12480, 4096, 4096, "gate_residual"),
                                        (12480, 4096, 4096, "lora"), (12480, 4096, 11008, "f32out"), (12480 - 300, 4096 + 40, 1024, "plain")])
def test_gemm_splitk_tail_matches_unsplit_and_fp32(M, N, K, kind, monkeypatch):
    """Tile counts that leave a thin last round on 256 CUs (784 = 3*256 + 16 tiles at the reference's 480p generation
    shapes; 48 x 17 = 816 = 3*256 + 48... is too fat and stays unsplit): the tail tiles are split along K, summed in fp32 and
    finished by the reduce kernel.
    Same epilogues as the unsplit kernel; results equal the unsplit ones up to the order of the fp32 partial sums (a
    handful of bf16 ulps on < 1 % of the elements) and the fp32 reference within the GEMM tolerance."""
    ops = _ops()
    from lcv_hip.lib import LCV_EPI_GATE_RESIDUAL
    a = _randn(M, K, seed=81).to(DEV); w = _randn(N, K, seed=82, scale=0.03).to(DEV); b = _randn(N, seed=83).to(DEV)
    kw = {}
    if kind == "gate_residual":
        T = 8
        kw = dict(epilogue=LCV_EPI_GATE_RESIDUAL, resid=_randn(M, N, seed=84).to(DEV),
                  mod=_randn(1, T, 3 * N, seed=85, dtype=torch.float32).to(DEV), gate_idx=2, rows_per_frame=M // T)
    if kind == "lora":
        kw = dict(a2=_randn(M, 64, seed=86).to(DEV), w2=_randn(N, 64, seed=87, scale=0.05).to(DEV))
    if kind == "f32out":
        kw = dict(out_f32=True)
    monkeypatch.setenv("LCV_GEMM_TILE", "9")                          # the 8-phase kernel (gemm4k.h, the default since round 4, has no split tail)
    monkeypatch.setenv("LCV_GEMM_SPLITK_TAIL", "0")
    ref = ops.gemm_nt(a, w, b, **kw)
    monkeypatch.delenv("LCV_GEMM_SPLITK_TAIL")
    out = ops.gemm_nt(a, w, b, **kw)
    assert torch.equal(ops.gemm_nt(a, w, b, **kw), out)            # deterministic (no atomics)
    d = (out.float() - ref.float()).abs()
    if kind != "f32out":                                            # bf16 outputs hide most of the fp32 reordering
        assert (d > 0).float().mean() < 0.01, (d > 0).float().mean()
    assert rel_l2(out, ref, bound=1.1e-5) < 1.1e-5, rel_l2(out, ref)
    if N == 4096:
        assert (d > 0).any()                                        # 784 tiles: the split path really ran
    else:
        assert not (d > 0).any()                                    # 48 x 17 = 816 tiles: tail of 48, left alone
    if kind in ("plain", "f32out"):
        rows = torch.randperm(M, generator=torch.Generator().manual_seed(2))[:64].to(DEV)
        exact = a[rows].float() @ w.float().t() + b.float()
        assert rel_l2(out[rows], exact, bound=2e-3) < 2e-3


def test_timestep_embedding_matches_oracle():
    """Sinusoidal timestep features (fp32) against the oracle's torch evaluation: arguments reach 1000 rad, so the absolute
    error of cos / sin is that of the fp32 product t * f (half an ulp of 1000 = 3e-5) on both sides."""
    ops, orc = _ops(), _orc()
    t = torch.tensor([0.0, 1.0, 37.5, 431.0, 612.0, 999.0, 1000.0], dtype=torch.float32)
    got = ops.timestep_embedding(t.to(DEV), 256).cpu()
    ref = orc.timestep_embedding(t, 256)
    assert got.shape == (7, 256) and (got - ref).abs().max() < 2e-4
    assert torch.equal(got[0], ref[0])                         # t = 0: cos 1, sin 0 exactly


def test_linear_f32_smallm():
    ops, orc = _ops(), _orc()
    import torch.nn.functional as F
    a = _randn(26, 512, seed=28, dtype=torch.float32); w = _randn(1536, 512, seed=29, scale=0.05); b = _randn(1536, seed=30)
    out = ops.linear_f32_smallm(a.to(DEV), w.to(DEV), b.to(DEV), act_in=1)
    ref = F.silu(a) @ w.float().t() + b.float()
    assert rel_l2(out, ref, bound=1.0e-6) < 1.0e-6


def test_patchify_unpatchify():
    ops, orc = _ops(), _orc()
    import torch.nn.functional as F
    B, Cin, T, H, W, C = 2, 16, 3, 8, 12, 128
    x = _randn(B, Cin, T, H, W, seed=31)
    wt = _randn(C, Cin, 1, 2, 2, seed=32, scale=0.1); bias = _randn(C, seed=33)
    tok = ops.patchify(x.to(DEV), 64)
    y = ops.gemm_nt(tok.view(-1, 64), wt.view(C, 64).to(DEV), bias.to(DEV)).view(B, -1, C)
    ref = orc.x_embedder({"x_embedder.proj.weight": wt, "x_embedder.proj.bias": bias}, x, (1, 2, 2), orc.bf16_round)
    assert rel_l2(y, ref, bound=1.0e-6) < 1.0e-6
    t = _randn(B, T * (H // 2) * (W // 2), 64, seed=34, dtype=torch.float32)
    out = ops.unpatchify(t.to(DEV), 16, T, H, W)
    assert torch.equal(out.cpu(), orc.unpatchify(t, T, H // 2, W // 2, (1, 2, 2), 16))


def test_denoise_glue_and_loss_pieces():
    ops = _ops()
    B, n = 2, 5000
    c = _randn(B, n, seed=35, dtype=torch.float32); u = _randn(B, n, seed=36, dtype=torch.float32)
    x = _randn(B, n, seed=37, dtype=torch.float32)
    xd = x.to(DEV).clone()
    ops.cfg_euler_step(c.to(DEV), u.to(DEV), xd, 4.0, -0.02, negate=True, zero_star=True)
    st = (c * u).sum(1, keepdim=True) / ((u * u).sum(1, keepdim=True) + 1e-8)
    v = u * st + 4.0 * (c - u * st)
    assert torch.allclose(xd.cpu(), x + (-0.02) * (-v), atol=1e-5, rtol=1e-5)
    # fm_noise: bit-exact with torch's (1-s)*x0 + s*eps in fp32 then .to(bf16)
    x0 = _randn(B, 16, 3, 4, 6, seed=38); eps = _randn(B, 16, 3, 4, 6, seed=39)
    sig = torch.tensor([0.3171, 0.9012], dtype=torch.float32)
    got = ops.fm_noise(x0.to(DEV), eps.to(DEV), sig.to(DEV))
    se = sig.view(B, 1, 1, 1, 1)
    assert torch.equal(got.cpu(), ((1.0 - se) * x0 + se * eps).to(BF16))
    # fm_mse
    pred = _randn(B, 16, 5, 4, 6, seed=40, dtype=torch.float32)
    loss, dpred = ops.fm_mse(pred.to(DEV), eps.to(DEV), x0.to(DEV), 2)
    tgt = (eps - x0).float()
    ref = torch.nn.functional.mse_loss(pred[:, :, 2:], tgt)
    assert abs(loss.item() - ref.item()) < 1e-5 * max(1.0, ref.item())
    gref = torch.zeros_like(pred); gref[:, :, 2:] = 2 * (pred[:, :, 2:] - tgt) / tgt.numel()
    assert torch.allclose(dpred.cpu(), gref, atol=1e-7, rtol=1e-5)


def test_gemm_8phase_race_screen_shapes_and_epilogues(monkeypatch):
    """A synchronisation edit makes a new schedule (cdna_hip_programming.md: "screen it for races over many runs at several
    sizes"): the 8-phase and the persistent 8-phase kernels against the one-barrier kernel, bit for bit, over even / odd
    K-tile counts (the tail of the LDS-DMA pipeline differs), with and without the rank-r K tile, ragged M / N edges, fewer
    and more tiles than CUs, and every fused epilogue."""
    ops = _ops()
    from lcv_hip.lib import LCV_EPI_GATE_RESIDUAL, LCV_EPI_GELU_TANH, LCV_EPI_NONE, LCV_EPI_SILU, LCV_EPI_SWIGLU
    g = torch.Generator().manual_seed(123)
    shapes = [(2048, 1024, 128, 0), (2049, 1030, 192, 0), (3000, 2050, 1088, 64), (256 * 19 + 7, 256 * 15, 256, 0),
              (256 * 30, 256 * 10 + 64, 320, 64), (4100, 1024, 2048, 0)]
    for (M, N, K, K2) in shapes:
        a = (torch.randn(M, K, generator=g)).to(BF16).to(DEV); w = (torch.randn(N, K, generator=g) * 0.05).to(BF16).to(DEV)
        b = torch.randn(N, generator=g).to(BF16).to(DEV)
        a2 = torch.randn(M, K2, generator=g).to(BF16).to(DEV) if K2 else None
        w2 = (torch.randn(N, K2, generator=g) * 0.05).to(BF16).to(DEV) if K2 else None
        resid = torch.randn(M, N, generator=g).to(BF16).to(DEV)
        mod = torch.randn(1, 3, 6 * N, generator=g).to(DEV)
        cases = [dict(epilogue=LCV_EPI_NONE), dict(epilogue=LCV_EPI_NONE, out_f32=True), dict(epilogue=LCV_EPI_GELU_TANH),
                 dict(epilogue=LCV_EPI_SILU),
                 dict(epilogue=LCV_EPI_GATE_RESIDUAL, resid=resid, mod=mod, gate_idx=2, rows_per_frame=(M + 2) // 3),
                 dict(epilogue=LCV_EPI_GATE_RESIDUAL, resid=resid)]
        if N % 64 == 0:
            cases.append(dict(epilogue=LCV_EPI_SWIGLU))
        for kw in cases:
            bias = None if kw["epilogue"] == LCV_EPI_SWIGLU else b
            monkeypatch.setenv("LCV_GEMM_TILE", "6")
            ref = ops.gemm_nt(a, w, bias, a2=a2, w2=w2, **kw)
            for tile in ("8", "9"):
                monkeypatch.setenv("LCV_GEMM_TILE", tile)
                for _ in range(2):
                    got = ops.gemm_nt(a, w, bias, a2=a2, w2=w2, **kw)
                    assert torch.equal(got, ref), (M, N, K, K2, kw["epilogue"], tile)
