"""CPU: internal consistency of the denoise-loop oracle (`oracle/pipeline_oracle.py`).  The loop is **parity unpinned**
(the upstream pipeline is absent offline); what CAN be checked without it: the two conditioning-frame
forms (pinned in the sequence at t = 0 / KV cache of one t = 0 pass) are the same function in exact arithmetic, the
schedule's end points and warp, the CFG-zero-star identities, and that one Euler step of the loop is the scheduler formula."""
import torch

from oracle import dit_oracle as D
from oracle import pipeline_oracle as PO


def _setup(seed=3):
    cfg = D.small_config(hidden_size=256, depth=2, num_heads=2, caption_channels=64)
    P = D.make_params(cfg, seed=seed, std=0.05)
    g = torch.Generator().manual_seed(seed)
    lat = torch.randn(1, 16, 4, 8, 8, generator=g)
    pe = torch.randn(1, 1, 12, 64, generator=g); pm = torch.ones(1, 12, dtype=torch.int64); pm[:, 9:] = 0
    ne = torch.randn(1, 1, 12, 64, generator=g); nm = torch.ones(1, 12, dtype=torch.int64)
    return cfg, P, lat, pe, pm, ne, nm


def test_kv_cached_and_pinned_conditioning_are_the_same_function_in_fp32():
    cfg, P, lat, pe, pm, ne, nm = _setup()
    a = PO.denoise(P, cfg, lat, pe, pm, ne, nm, num_cond_latents=2, num_inference_steps=3, use_kv_cache=True, bf16=False)
    b = PO.denoise(P, cfg, lat, pe, pm, ne, nm, num_cond_latents=2, num_inference_steps=3, use_kv_cache=False, bf16=False)
    assert torch.equal(a[:, :, :2], lat[:, :, :2]) and torch.equal(b[:, :, :2], lat[:, :, :2])   # cond frames untouched
    assert (a - b).abs().max().item() < 2e-4 * b.abs().max().item()
    assert not torch.allclose(a[:, :, 2:], lat[:, :, 2:])


def test_schedule_end_points_shift_and_step_formula():
    ts, sig = PO.sigma_grid(50)
    assert len(ts) == 50 and len(sig) == 51 and sig[0] == 1.0 and abs(float(sig[49]) - 0.001) < 1e-9 and sig[50] == 0.0
    assert torch.equal(ts, sig[:50] * 1000)
    ts7, sig7 = PO.sigma_grid(10, shift=7.0)
    base = torch.linspace(1, 0.001, 10)
    assert torch.allclose(sig7[:10], 7 * base / (1 + 6 * base)) and sig7[0] == 1.0
    x = torch.randn(2, 5); v = torch.randn(2, 5)
    assert torch.allclose(PO.euler_update(x, v, -0.02), x + 0.02 * v)
    assert torch.allclose(PO.euler_update(x, v, -0.02, negate=False), x - 0.02 * v)


def test_cfg_zero_star_identities():
    g = torch.Generator().manual_seed(1)
    c = torch.randn(1, 16, 3, 4, 4, generator=g); u = torch.randn(1, 16, 3, 4, 4, generator=g)
    # guidance 1 returns the conditional prediction; c == u returns it for every guidance; c orthogonal to u: st = 0
    assert torch.allclose(PO.cfg_zero_star(c, u, 1.0), c, atol=1e-6)
    assert torch.allclose(PO.cfg_zero_star(c, c, 4.0), c, atol=1e-5)
    u_orth = u - (c * u).sum() / (c * c).sum() * c
    assert torch.allclose(PO.cfg_zero_star(c, u_orth, 4.0), 4.0 * c, atol=1e-4)
    # scale invariance in u: the unconditional branch only contributes its direction
    assert torch.allclose(PO.cfg_zero_star(c, 3.0 * u, 4.0), PO.cfg_zero_star(c, u, 4.0), atol=1e-4)


def test_loop_without_cfg_and_without_cond_is_plain_euler_on_the_dit():
    cfg, P, lat, pe, pm, _, _ = _setup(5)
    seen = []
    out = PO.denoise(P, cfg, lat, pe, pm, None, None, num_cond_latents=0, num_inference_steps=2, guidance_scale=1.0,
                     bf16=True, step_callback=lambda i, x: seen.append(x.clone()))
    ts, sig = PO.sigma_grid(2)
    x = lat.clone()
    for i in range(2):
        t = D.bf16_round(torch.full((1, 4), float(ts[i])))
        pred = D.dit_forward(P, cfg, D.bf16_round(x), t, pe, pm, 0, bf16=True)
        x = x + (float(sig[i + 1]) - float(sig[i])) * (-pred)
        assert torch.allclose(seen[i], x, atol=1e-6)
    assert torch.allclose(out, x, atol=1e-6)
