"""smoke(): one tiny invocation of the hot path on cuda:0, checked against the CPU oracle.
(The oracle is imported here as the checker only — __graft_entry__.smoke() is one of the three places allowed to.)"""
import torch


def run_smoke():
    from oracle import dit_oracle as orc
    from longcat_video.modules.longcat_video_dit import LongCatVideoTransformer3DModel
    from longcat_video.modules.scheduling_flow_match_euler_discrete import FlowMatchEulerDiscreteScheduler
    from longcat_video.pipeline_longcat_video import LongCatVideoPipeline
    BF16 = torch.bfloat16
    cfg = orc.small_config(hidden_size=256, depth=2, num_heads=2, caption_channels=64)
    P = orc.make_params(cfg, seed=11, std=0.05)
    dit = LongCatVideoTransformer3DModel(device="cuda:0", dtype=BF16, hidden_size=256, depth=2, num_heads=2,
                                         caption_channels=64, adaln_tembed_dim=64).eval()
    dit.load_state_dict(P, strict=False)
    g = torch.Generator().manual_seed(0)
    hs = torch.randn(1, 16, 3, 8, 8, generator=g).to(BF16)
    y = torch.randn(1, 1, 16, 64, generator=g).to(BF16)
    mask = torch.zeros(1, 16, dtype=torch.int64); mask[:, :9] = 1
    ts = torch.zeros(1, 3); ts[:, 1:] = 640.0
    with torch.no_grad():
        got = dit(hs.cuda(), ts.to(BF16).cuda(), y.cuda(), mask.cuda(), num_cond_latents=1)
    ref = orc.dit_forward(P, cfg, hs, ts.to(BF16), y, mask, 1, bf16=True)
    err = (torch.linalg.vector_norm(got.cpu() - ref) / torch.linalg.vector_norm(ref)).item()
    assert err < 1e-2, f"DiT forward differs from the oracle: rel-L2 {err:.3e}"
    # two denoise steps through the pipeline (conditioning-frame KV cache, CFG-zero-star, fused Euler update) vs the loop oracle
    from oracle import pipeline_oracle as porc
    pipe = LongCatVideoPipeline(scheduler=FlowMatchEulerDiscreteScheduler(), dit=dit)
    lat = torch.randn(1, 16, 3, 8, 8, generator=g)
    ne = torch.randn(1, 1, 16, 64, generator=g).to(BF16)
    out = pipe.denoise(lat.cuda(), y.cuda(), mask.cuda(), ne.cuda(), mask.cuda(), num_cond_latents=1,
                       num_inference_steps=2, guidance_scale=4.0, use_kv_cache=True)
    assert out.shape == lat.shape and torch.isfinite(out).all()
    want = porc.denoise(P, cfg, lat, y, mask, ne, mask, num_cond_latents=1, num_inference_steps=2, guidance_scale=4.0,
                        use_kv_cache=True, bf16=True)
    upd = (torch.linalg.vector_norm((out.cpu() - lat) - (want - lat)) / torch.linalg.vector_norm(want - lat)).item()
    assert upd < 3e-2, f"2-step denoise update differs from the oracle: rel-L2 {upd:.3e}"
    print(f"smoke: DiT fwd rel-L2 vs oracle {err:.2e}; 2-step KV-cached CFG denoise update rel-L2 vs oracle {upd:.2e}")
