"""CPU: the augmentation variants of the TTA clip (tta/augment.py) against a fixture minted from the reference's own functions
(tests/golden/make_golden_augment.py -> augment.pt: delta_experiment/scripts/common.py:1161-1314).  Flip, speed indices, names,
order, the zoom factor and the random-angle draw are pinned; the rotation warp itself is torchvision's in the reference and is
checked through properties (parity unpinned for it: tta/augment.py header)."""
import math
from pathlib import Path

import pytest
import torch

GOLD = torch.load(Path(__file__).parent / "golden" / "augment.pt")


def test_parse_speed_factors_and_rotation_scale_match_the_reference():
    from tta import augment as A
    for raw, want in GOLD["parse_speed_factors"]:
        assert A.parse_speed_factors(raw) == want
    for h, w, d, want in GOLD["rotation_scale"]:
        assert A._rotation_scale(h, w, d) == want            # same fp32 sin / cos, same arithmetic: exact


def test_pixel_variants_flip_and_speed_match_the_reference_bitwise():
    from tta import augment as A
    clip = GOLD["clip"]
    for case in GOLD["pixel_variants"]:
        got = A.build_augmented_pixel_variants(clip, **case["kw"])
        assert [v["name"] for v in got] == case["names"]
        for v, want in zip(got, case["frames"]):
            assert v["pixel_frames"].shape == want.shape and torch.equal(v["pixel_frames"], want), v["name"]


def test_random_rotation_draws_consume_the_rng_like_the_reference(monkeypatch):
    from tta import augment as A
    seen = []
    monkeypatch.setattr(A, "rotate_clip", lambda px, deg, zoom=False: (seen.append(deg), px)[1])
    clip = GOLD["clip"]
    for d in GOLD["angle_draws"]:
        rmin, rmax, cnt, step = d["args"]
        seen.clear()
        torch.manual_seed(d["seed"])
        vs = A.build_augmented_pixel_variants(clip, rotate_deg=0.0, rotate_random_min=rmin, rotate_random_max=rmax,
                                              rotate_random_count=cnt, rotate_random_step=step)
        kept = [a for a in d["angles"] if abs(a) >= 1e-6]
        assert seen == [float(a) for a in kept]
        assert [v["name"] for v in vs] == ["orig"] + [f"rotate_rand_{float(a):+.1f}" for a in kept]
    # fixed rotations: -deg then +deg, named like the reference
    seen.clear()
    vs = A.build_augmented_pixel_variants(clip, rotate_deg=10.0, rotate_random_count=0)
    assert seen == [-10.0, 10.0] and [v["name"] for v in vs] == ["orig", "rotate_-10.0", "rotate_+10.0"]


def test_rotation_warp_properties():
    from tta import augment as A
    g = torch.Generator().manual_seed(3)
    clip = torch.rand(1, 3, 4, 16, 16, generator=g)
    assert torch.allclose(A.rotate_clip(clip, 0.0), clip, atol=1e-6)                        # identity
    r90 = A.rotate_clip(clip, 90.0)                                                         # a square frame, quarter turn:
    cw, ccw = torch.rot90(clip, -1, dims=(3, 4)), torch.rot90(clip, 1, dims=(3, 4))          # exact up to interpolation noise
    assert min((r90 - cw).abs().max().item(), (r90 - ccw).abs().max().item()) < 1e-4
    assert (r90 - cw).abs().max().item() < 1e-4        # torchvision's `affine`: a positive angle turns the picture clockwise
    back = A.rotate_clip(A.rotate_clip(clip, 90.0), -90.0)
    assert torch.allclose(back, clip, atol=1e-4)
    # a zoomed rotation of a constant frame stays constant (no empty corners), an unzoomed one does not
    ones = torch.ones(1, 3, 2, 24, 40)
    z = A.rotate_clip(ones, 12.0, zoom=True)            # (the rotated corners land ON the frame edge, half a pixel outside the
    inner = z[..., 1:-1, 1:-1]                           # outermost pixel centres: only the four corner pixels see the padding)
    assert torch.allclose(inner, torch.ones_like(inner), atol=1e-5) and z.min().item() > 0.5
    assert (z < 0.999).sum().item() <= 4 * 3 * 2
    assert A.rotate_clip(ones, 12.0, zoom=False).min().item() < 0.01
    # linear in the image, frames independent
    a, b = clip, torch.rand(1, 3, 4, 16, 16, generator=g)
    assert torch.allclose(A.rotate_clip(a + 2 * b, 7.0, zoom=True), A.rotate_clip(a, 7.0, zoom=True) + 2 * A.rotate_clip(b, 7.0, zoom=True), atol=1e-5)
    assert torch.equal(A.rotate_clip(clip, 7.0)[:, :, 1:2], A.rotate_clip(clip[:, :, 1:2], 7.0))


def test_train_variants_are_cut_to_the_training_window_and_short_clips_skipped():
    from types import SimpleNamespace
    from tta import augment as A

    class FakeVae:            # the latent clip of a pixel clip: one latent frame per 4 pixel frames after the first, value = mean
        dtype = torch.float32

    calls = []

    def fake_encode(vae, px):
        calls.append(tuple(px.shape))
        T = 1 + (px.shape[2] - 1) // 4
        return torch.stack([px[:, :1, 0 if t == 0 else 4 * t - 3:4 * t + 1].mean(dim=(1, 2, 3, 4)) for t in range(T)], dim=1).view(1, 1, T, 1, 1)
    import tta.augment as mod
    orig = mod._encode
    mod._encode = fake_encode
    try:
        px = torch.rand(1, 3, 29, 4, 6)
        cond = torch.zeros(1, 1, 4, 1, 1); train = torch.zeros(1, 1, 3, 1, 1)
        args = SimpleNamespace(aug_flip=True, aug_rotate_deg=0.0, aug_rotate_random_min=5.0, aug_rotate_random_max=15.0,
                               aug_rotate_random_count=0, aug_rotate_random_step=1.0, aug_rotate_zoom=True, aug_speed_factors="0.5,2.0")
        vs = A.build_train_latents_variants(FakeVae(), px, cond, train, args)
    finally:
        mod._encode = orig
    assert [v["name"] for v in vs] == ["orig", "flip_h", "slow_2x"]          # speed_2x: 15 frames -> 4 latents < 7: skipped
    assert vs[0]["latents"] is train and all(v["latents"].shape[2] == 3 for v in vs)
    assert len(calls) == 3
