"""GPU: `LongCatVideoPipeline.generate_vc` end to end on tiny synthetic models — pixel frames -> HIP VAE encode -> clean
conditioning latents -> KV-cached CFG denoise -> HIP VAE decode — i.e. the call the reference's runners make
(delta_experiment/scripts/common.py:566-611).  Checks the output contract, that the KV-cached path and the pinned-in-sequence
path agree, and that the conditioning latents really are the normalised posterior mode of the encoder."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
BF16 = torch.bfloat16


def _pipe():
    from longcat_video import pipeline_longcat_video as PL
    from longcat_video.modules.autoencoder_kl_wan import AutoencoderKLWan
    from longcat_video.modules.longcat_video_dit import LongCatVideoTransformer3DModel
    from longcat_video.modules.scheduling_flow_match_euler_discrete import FlowMatchEulerDiscreteScheduler
    dit = LongCatVideoTransformer3DModel(device="cuda", dtype=BF16, depth=2, hidden_size=256, num_heads=2,
                                         caption_channels=64).init_synthetic_(11)
    vae = AutoencoderKLWan(base_dim=16, z_dim=16, device="cuda", dtype=BF16).init_synthetic_(12)
    pipe = PL.LongCatVideoPipeline(vae=vae, scheduler=FlowMatchEulerDiscreteScheduler(), dit=dit)
    pipe.device = torch.device("cuda")
    return PL, pipe


def test_generate_vc_end_to_end(monkeypatch):
    PL, pipe = _pipe()
    monkeypatch.setitem(PL.RESOLUTIONS, "tiny", (32, 48))
    g = torch.Generator().manual_seed(5)
    video = (torch.rand(9, 32, 48, 3, generator=g) * 255).to(torch.uint8).numpy()
    pe = torch.randn(1, 1, 16, 64, generator=g).to(BF16).cuda(); pm = torch.ones(1, 16, dtype=torch.int64).cuda(); pm[:, 11:] = 0
    ne = torch.randn(1, 1, 16, 64, generator=g).to(BF16).cuda(); nm = torch.ones(1, 16, dtype=torch.int64).cuda()
    outs = {}
    for use_kv in (True, False):
        gen = torch.Generator(device="cuda").manual_seed(9)
        outs[use_kv] = pipe.generate_vc(video, resolution="tiny", num_frames=17, num_cond_frames=5, num_inference_steps=3,
                                        guidance_scale=4.0, generator=gen, use_kv_cache=use_kv, prompt_embeds=pe, prompt_mask=pm,
                                        negative_embeds=ne, negative_mask=nm)[0]
    a, b = outs[True], outs[False]
    assert isinstance(a, np.ndarray) and a.shape == (17, 32, 48, 3) and a.dtype == np.float32
    assert np.isfinite(a).all() and a.min() >= 0.0 and a.max() <= 1.0
    assert np.abs(a - b).mean() < 2e-2   # cached vs pinned conditioning: same maths, different kernel launches / rounding
    # the conditioning latents are the normalised posterior mode of the last num_cond_frames frames
    gen = torch.Generator(device="cuda").manual_seed(9)
    lat = pipe.generate_vc(video, resolution="tiny", num_frames=17, num_cond_frames=5, num_inference_steps=3, generator=gen,
                           prompt_embeds=pe, prompt_mask=pm, negative_embeds=ne, negative_mask=nm, output_type="latent")[0]
    frames = pipe._frames_to_tensor(video, 32, 48)[:, :, -5:]
    z = PL.retrieve_latents(pipe.vae.encode(frames.to(BF16)), sample_mode="argmax").float()
    mean = torch.tensor(pipe.vae.config.latents_mean, device="cuda").view(1, -1, 1, 1, 1)
    std = torch.tensor(pipe.vae.config.latents_std, device="cuda").view(1, -1, 1, 1, 1)
    assert lat.shape == (1, 16, 5, 4, 6) and torch.allclose(lat[:, :, :2].float(), (z - mean) / std, atol=1e-5)


def test_common_helpers_mirror_the_reference_contract():
    """tta/common.py: the same-named counterparts of delta_experiment/scripts/common.py:158-255, 566-611 on synthetic
    components — latent normalisation round trip in the latents' dtype, decode contract ((v + 1) / 2, clamped, [B,3,N,H,W]),
    prompt encoding through the HIP UMT5 with a stand-in tokenizer, and the continuation's frame rounding."""
    from types import SimpleNamespace
    from longcat_video.modules.autoencoder_kl_wan import AutoencoderKLWan
    from longcat_video.modules.longcat_video_dit import LongCatVideoTransformer3DModel
    from longcat_video.modules.scheduling_flow_match_euler_discrete import FlowMatchEulerDiscreteScheduler
    from longcat_video.modules.umt5_encoder import UMT5EncoderModel
    from longcat_video.pipeline_longcat_video import LongCatVideoPipeline
    from tta import common as C
    vae = AutoencoderKLWan(device="cuda").init_synthetic_()
    z = torch.randn(1, 16, 3, 8, 12, device="cuda").to(BF16)
    n = C.normalize_latents(vae, z)
    assert n.dtype == BF16 and n.shape == z.shape
    mean = torch.tensor(vae.config.latents_mean).view(1, 16, 1, 1, 1).to("cuda", BF16)
    inv = 1.0 / torch.tensor(vae.config.latents_std).view(1, 16, 1, 1, 1).to("cuda", BF16)
    assert torch.equal(n, (z - mean) * inv) and torch.equal(C.denormalize_latents(vae, n), n / inv + mean)
    frames = torch.rand(1, 3, 5, 64, 96, device="cuda").to(BF16) * 2 - 1
    torch.manual_seed(5)
    lat = C.encode_video(vae, frames, normalize=True)
    assert lat.shape == (1, 16, 2, 8, 12)
    vid = C.decode_latents(vae, lat)
    assert vid.shape == (1, 3, 5, 64, 96) and float(vid.min()) >= 0.0 and float(vid.max()) <= 1.0
    raw = vae.decode(C.denormalize_latents(vae, lat).to(vae.dtype), return_dict=False)[0]
    assert torch.equal(vid, ((raw + 1.0) / 2.0).clamp(0, 1))

    class Tok:
        def __call__(self, texts, **kw):
            L = kw["max_length"]
            ids = torch.zeros(1, L, dtype=torch.long); ids[0, :6] = torch.arange(3, 9)
            m = torch.zeros(1, L, dtype=torch.long); m[0, :6] = 1
            return SimpleNamespace(input_ids=ids, attention_mask=m)
    enc = UMT5EncoderModel(device="cuda", vocab_size=64, d_model=64, d_kv=64, d_ff=128, num_layers=1, num_heads=1).init_synthetic_()
    emb, mask = C.encode_prompt(Tok(), enc, "a red kite", device="cuda")
    assert emb.shape == (1, 1, 512, 64) and emb.dtype == BF16 and int(mask.sum()) == 6

    dit = LongCatVideoTransformer3DModel(device="cuda", dtype=BF16, depth=1, hidden_size=256, num_heads=2, caption_channels=64).init_synthetic_(3)
    pipe = LongCatVideoPipeline(vae=vae, scheduler=FlowMatchEulerDiscreteScheduler(), dit=dit)
    pipe.device = torch.device("cuda")
    video_u8 = (torch.rand(6, 480, 832, 3) * 255).to(torch.uint8).numpy()
    pe = torch.randn(1, 1, 512, 64, device="cuda").to(BF16); pm = torch.zeros(1, 512, dtype=torch.int64, device="cuda"); pm[:, :9] = 1
    out = C.generate_video_continuation(pipe, video_u8, "a red kite", num_cond_frames=5, num_frames=11, num_inference_steps=2,
                                        guidance_scale=4.0, seed=1, device="cuda", prompt_embeds=pe, prompt_mask=pm,
                                        negative_embeds=torch.zeros_like(pe), negative_mask=pm)
    # num_frames 11 -> num_frames_valid 13 (common.py:589-593): 4 latent frames, 13 decoded frames
    assert out.shape == (13, 480, 832, 3) and out.dtype.kind == "f" and 0.0 <= float(out.min()) and float(out.max()) <= 1.0
