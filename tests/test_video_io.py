"""CPU: the raw-video leg in front of the hot path (tta/video_io.py) against what the REFERENCE's decoders produced over the same
seeded clips (tests/golden/video_io.pt, make_video_io_golden.py; PyAV is absent from this image, tests/fake_av.py stands in with the
handful of calls both sides make), and the blob `load_entry` builds from a raw video with stand-in VAE / tokenizer / text encoder."""
import sys
import types
from pathlib import Path

import numpy as np
import pytest
import torch

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT / "longcat-video-tta_amd")); sys.path.insert(0, str(ROOT / "tests"))
import fake_av  # noqa: E402

FIX = torch.load(ROOT / "tests" / "golden" / "video_io.pt")


@pytest.fixture()
def av(monkeypatch):
    monkeypatch.setitem(sys.modules, "av", fake_av.install())


def test_frame_windows_equal_the_reference_bit_for_bit(av):
    from tta import video_io as V
    for c in FIX["frames"]:
        got = V.load_video_frames(c["path"], c["num_frames"], height=c["height"], width=c["width"], start_frame=c["start_frame"])
        assert got.dtype == torch.float32 and torch.equal(got, c["pixels"]), c["path"]


def test_ground_truth_frames_reproduce_the_reference_psnr(av):
    """The reference keeps its ground-truth frames inside `evaluate_generation_metrics`; what it returned for a seeded clip pins them:
    the same PSNR from OUR frames (decode from gen_start_frame, PIL LANCZOS to the output size) to 1e-6."""
    from tta import video_io as V
    for key, n_expect in (("gt", 4), ("gt_short", 2)):
        c = dict(FIX["gt"], **FIX[key])
        gen = np.random.RandomState(c["gen_seed"]).rand(*c["gen_shape"]).astype(np.float32)[c["num_cond_frames"]:]
        gt = V.load_ground_truth_frames(c["path"], c["gen_start_frame"], c["num_gen_frames"], c["gen_shape"][1], c["gen_shape"][2])
        assert gt.dtype == np.uint8 and gt.shape == (n_expect,) + tuple(c["gen_shape"][1:])
        gtf = gt.astype(np.float64) / 255.0
        psnr = float(np.mean([10.0 * np.log10(1.0 / np.mean((gen[i].astype(np.float32) - gtf[i].astype(np.float32)) ** 2)) for i in range(n_expect)]))
        assert abs(psnr - c["psnr"]) < 1e-5, (key, psnr, c["psnr"])
    assert V.load_ground_truth_frames("fake://1/3/8x8", 5, 2, 8, 8) is None


def test_without_pyav_the_error_names_the_alternative(monkeypatch):
    from tta import video_io as V
    monkeypatch.setitem(sys.modules, "av", None)
    with pytest.raises(ImportError, match="pre-encode"):
        V.load_video_frames("fake://1/3/8x8", 2)


class _VAE:
    dtype = torch.float32
    config = types.SimpleNamespace(z_dim=16, latents_mean=[0.1] * 16, latents_std=[2.0] * 16)

    def encode(self, x):                        # [1, 3, T, H, W] -> a 16-channel, 4x / 8x / 8x smaller "posterior"
        T = 1 + (x.shape[2] - 1) // 4
        z = torch.nn.functional.adaptive_avg_pool3d(x.float(), (T, x.shape[3] // 8, x.shape[4] // 8)).mean(1, keepdim=True).repeat(1, 16, 1, 1, 1)
        return types.SimpleNamespace(latent_dist=types.SimpleNamespace(sample=lambda generator=None: z, mode=lambda: z))


class _Tok:
    def __call__(self, texts, **kw):
        assert kw["padding"] == "max_length" and kw["max_length"] == 512 and kw["truncation"] and kw["add_special_tokens"]
        n = min(512, len(texts[0].split()) + 1)
        ids = torch.zeros(1, 512, dtype=torch.int64); ids[0, :n] = torch.arange(1, n + 1)
        mask = torch.zeros(1, 512, dtype=torch.int64); mask[0, :n] = 1
        return types.SimpleNamespace(input_ids=ids, attention_mask=mask)


class _Enc:
    def __call__(self, ids, mask):
        return types.SimpleNamespace(last_hidden_state=(ids.float()[..., None] * torch.ones(64)) * mask[..., None])


def test_load_entry_builds_the_whole_blob_from_a_raw_video(av):
    from tta import runner_common as R
    from tta import video_io as V
    pipe = types.SimpleNamespace(vae=_VAE(), tokenizer=_Tok(), text_encoder=_Enc())
    args = types.SimpleNamespace(resolution="480p", tta_total_frames=13, tta_context_frames=5, gen_start_frame=20, num_cond_frames=5,
                                 num_frames=9, skip_generation=False)
    entry = {"kind": "video", "name": "v", "path": "fake://31/30/24x40", "caption": "a person rides a bike", "class_name": "Biking"}
    blob = R.load_entry(entry, args, None, "cpu", pipe=pipe)
    # TTA window = 13 frames ending at frame 20, at the reference's hard-coded 480 x 832; latents normalised, bf16
    want_pix = V.load_video_frames(entry["path"], 13, 480, 832, start_frame=7).to(torch.bfloat16)
    assert torch.equal(blob["pixel_frames"], want_pix) and blob["latents"].dtype == torch.bfloat16
    z = pipe.vae.encode(want_pix).latent_dist.sample()
    assert blob["latents"].shape == (1, 16, 4, 60, 104) and torch.allclose(blob["latents"].float(), ((z - 0.1) / 2.0).to(torch.bfloat16).float())
    assert blob["prompt_embeds"].shape == (1, 1, 512, 64) and int(blob["prompt_mask"].sum()) == 6 and int(blob["negative_mask"].sum()) == 1
    assert blob["caption"] == entry["caption"]
    # conditioning frames of the continuation: 5 frames ending at frame 20, quantised as the reference quantises them
    cond = V.frames_to_uint8(V.load_video_frames(entry["path"], 5, 480, 832, start_frame=15))
    assert blob["cond_frames"].dtype == torch.uint8 and torch.equal(blob["cond_frames"], cond)
    # ground truth: frames 20 .. 23 at the output size
    assert blob["gt_frames"].shape == (4, 480, 832, 3) and blob["gt_frames"].dtype == torch.uint8
    assert np.array_equal(blob["gt_frames"].numpy(), V.load_ground_truth_frames(entry["path"], 20, 4, 480, 832))
    # 720p output: conditioning frames and ground truth at 720 x 1280, the TTA window still at 480 x 832 (Appendix B)
    args.resolution = "720p"
    b2 = R.load_entry(entry, args, None, "cpu", pipe=pipe)
    assert b2["cond_frames"].shape == (5, 720, 1280, 3) and b2["gt_frames"].shape == (4, 720, 1280, 3) and b2["pixel_frames"].shape[-2:] == (480, 832)
    # a pipeline without a text encoder says what is missing
    with pytest.raises(RuntimeError, match="text_encoder"):
        R.load_entry(entry, args, None, "cpu", pipe=types.SimpleNamespace(vae=_VAE(), tokenizer=_Tok(), text_encoder=None))
