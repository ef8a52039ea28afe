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
