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


def test_encoder_full_equals_chunked_and_contract():
    """Streaming encode (frame 0 alone, then 4 frames per call with per-conv caches and the cached-last-frame temporal
    downsample) == whole-sequence encode (what the HIP encoder implements); shape contract [1,3,1+4k,8h,8w] -> [1,z,1+k,h,w]."""
    cfg = V.default_config(base_dim=8, z_dim=4)
    P = {k: v.float() for k, v in V.make_encoder_params(cfg, seed=3, dtype=torch.float32).items()}
    g = torch.Generator().manual_seed(4)
    for k in (0, 1, 3):
        video = torch.rand(1, 3, 1 + 4 * k, 16, 24, generator=g) * 2 - 1
        full = V.encode_full(P, cfg, video)
        chunked = V.encode_chunked(P, cfg, video)
        assert full.shape == (1, 4, 1 + k, 2, 3)
        assert torch.allclose(full, chunked, atol=1e-5, rtol=1e-5), (k, (full - chunked).abs().max())
    # causality: later pixel frames cannot change earlier latent frames
    video = torch.rand(1, 3, 9, 16, 24, generator=g) * 2 - 1
    a = V.encode_full(P, cfg, video)
    v2 = video.clone(); v2[:, :, 5:] += 0.5
    b = V.encode_full(P, cfg, v2)
    assert torch.equal(a[:, :, :2], b[:, :, :2]) and not torch.equal(a[:, :, 2:], b[:, :, 2:])


def test_encoder_plan_matches_wan21():
    dims, plan = V.encoder_plan(V.default_config())
    assert dims == [96, 96, 192, 384, 384]
    assert [p[0] for p in plan] == ["res", "res", "down2d", "res", "res", "down3d", "res", "res", "down3d", "res", "res"]
    assert plan[3] == ("res", 96, 192) and plan[6] == ("res", 192, 384) and plan[9] == ("res", 384, 384)
