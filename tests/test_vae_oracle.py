"""CPU: the two formulations of the causal VAE decoder in the oracle agree — streaming one latent frame at a time with
per-conv feature caches (the upstream module's way) vs the whole sequence at once with causal zero padding (what the HIP
decoder implements) — and the output contract of delta_experiment/scripts/common.py:209-221 holds."""
import torch

from oracle import vae_oracle as V


def test_full_equals_chunked_and_contract():
    cfg = V.default_config(base_dim=8, z_dim=4)
    P = {k: v.float() for k, v in V.make_params(cfg, seed=1, dtype=torch.float32).items()}
    g = torch.Generator().manual_seed(2)
    for T in (1, 2, 4):
        z = torch.randn(1, 4, T, 3, 5, generator=g)
        full = V.decode_full(P, cfg, z)
        chunked = V.decode_chunked(P, cfg, z)
        assert full.shape == (1, 3, 1 + 4 * (T - 1), 24, 40)
        assert torch.allclose(full, chunked, atol=1e-5, rtol=1e-5), (T, (full - chunked).abs().max())
        assert full.min() >= -1 and full.max() <= 1
    # causality: later latent frames cannot change earlier output frames
    z = torch.randn(1, 4, 3, 3, 5, generator=g)
    a = V.decode_full(P, cfg, z)
    z2 = z.clone(); z2[:, :, 2] += 1.0
    b = V.decode_full(P, cfg, z2)
    assert torch.equal(a[:, :, :5], b[:, :, :5]) and not torch.equal(a[:, :, 5:], b[:, :, 5:])


def test_decoder_plan_matches_wan21():
    dims, plan = V.decoder_plan(V.default_config())
    assert dims == [384, 384, 384, 192, 96]
    assert plan == [(384, 384, 3, "upsample3d"), (192, 384, 3, "upsample3d"), (192, 192, 3, "upsample2d"),
                    (96, 96, 3, None)]
