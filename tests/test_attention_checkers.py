"""CPU: the attention checkers of oracle/dit_oracle.py that restate the kernels' ROUNDING POINTS (what the `-m gpu` tests hold the HIP
kernels to at 1e-4) are themselves the same function as the plain fp32 definitions, to the bf16 rounding they add - and the
deferred-max rule behaves as csrc/attn_fwd.hip:284-345 states it (decision per 32-row group, rows of other groups untouched)."""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from oracle import dit_oracle as orc  # noqa: E402


def _rel(a, b):
    return ((a.float() - b.float()).norm() / b.float().norm()).item()


def _inputs(Nq, Nk, seed=0, H=2, D=128):
    g = torch.Generator().manual_seed(seed)
    mk = lambda n: torch.randn(1, H, n, D, generator=g).bfloat16()
    return mk(Nq), mk(Nk), mk(Nk), mk(Nq)


def test_forward_restatement_is_the_softmax_up_to_bf16_rounding():
    q, k, v, _ = _inputs(70, 333)
    scale = 128 ** -0.5
    ref = orc.sdpa(q, k, v, scale)
    assert _rel(orc.sdpa_at_kernel_rounding(q, k, v, scale), ref) < 3e-3
    # one tile: nothing is deferred, the restatement is bf16((bf16(exp(s - max)) v) / sum) exactly
    q1, k1, v1, _ = _inputs(33, 50, seed=1)
    s = (q1.float() @ k1.float().transpose(-1, -2)) * (scale * 1.4426950408889634)
    p = torch.exp2(s - s.amax(-1, keepdim=True))
    direct = orc.bf16_round((orc.bf16_round(p) @ v1.float()) / p.sum(-1, keepdim=True))
    assert torch.equal(orc.sdpa_at_kernel_rounding(q1, k1, v1, scale), direct)


def test_deferred_max_is_decided_per_32_row_group():
    q, k, v, _ = _inputs(96, 256, seed=2, H=1)
    k[0, 0, 200] = (q[0, 0, 40].float() * 40.0).bfloat16()       # query 40 (group 1) meets a huge score in key tile 3
    scale = 128 ** -0.5
    lazy = orc.sdpa_at_kernel_rounding(q, k, v, scale)            # threshold 2^6: group 1 rescales at tile 3, groups 0 and 2 do not
    eager = orc.sdpa_at_kernel_rounding(q, k, v, scale, thr=-1.0)  # every tile raises every row's max: plain flash attention
    assert _rel(lazy, orc.sdpa(q, k, v, scale)) < 3e-3 and _rel(eager, orc.sdpa(q, k, v, scale)) < 3e-3
    assert not torch.equal(lazy, eager)                           # P was rounded against different maxima somewhere ...
    shifted = orc.sdpa_at_kernel_rounding(q[:, :, 32:], k, v, scale, row0=32)
    assert torch.equal(shifted, lazy[:, :, 32:])                  # ... and a slice of rows, with its offset, is the same function


def test_backward_checker_is_autograd_up_to_bf16_rounding_in_both_forms():
    q, k, v, g = _inputs(70, 200, seed=3)
    scale = 128 ** -0.5
    qf, kf, vf = (t.float().requires_grad_(True) for t in (q, k, v))
    (torch.softmax(qf @ kf.transpose(-1, -2) * scale, -1) @ vf).backward(g.float())
    for inside in (True, False):
        dq, dk, dv = orc.sdpa_backward_at_kernel_rounding(q, k, v, g, scale, scale_inside=inside)
        assert max(_rel(dq, qf.grad), _rel(dk, kf.grad), _rel(dv, vf.grad)) < 4e-3
    # the forward's outputs as inputs (what the kernels are handed): same function
    o = orc.bf16_round(orc.sdpa(q, k, v, scale))
    lse = torch.logsumexp(q.float() @ k.float().transpose(-1, -2) * scale, -1)
    dq2, dk2, dv2 = orc.sdpa_backward_at_kernel_rounding(q, k, v, g, scale, o=o, lse=lse)
    assert max(_rel(dq2, qf.grad), _rel(dk2, kf.grad), _rel(dv2, vf.grad)) < 4e-3
