"""hipGraph replay of the denoise step vs the eager loop: K1 (1 280 tokens) and the reference's 480p KV-cached point (6 240 noise tokens),
48 blocks, CFG, 20 steps each."""
import os, sys, time, torch
sys.path.insert(0, "longcat-video-tta_amd"); sys.path.insert(0, ".")
from longcat_video.modules.longcat_video_dit import LongCatVideoTransformer3DModel
from longcat_video.modules.scheduling_flow_match_euler_discrete import FlowMatchEulerDiscreteScheduler
from longcat_video.pipeline_longcat_video import LongCatVideoPipeline
dev = "cuda"; bf = torch.bfloat16
dit = LongCatVideoTransformer3DModel(device=dev, dtype=bf, depth=48).eval().init_synthetic_()
for p in dit.parameters(): p.requires_grad = False
pipe = LongCatVideoPipeline(scheduler=FlowMatchEulerDiscreteScheduler(), dit=dit); pipe.device = torch.device(dev)
g = torch.Generator(device=dev).manual_seed(1)
pe = torch.randn(1, 1, 512, 4096, device=dev, generator=g).to(bf); ne = torch.randn(1, 1, 512, 4096, device=dev, generator=g).to(bf)
pm = torch.zeros(1, 512, dtype=torch.int64, device=dev); pm[:, :77] = 1; nm = pm.clone()
for name, shape, ncond in (("K1 16x256x256", (1, 16, 5, 32, 32), 0), ("480p 14c+14g, KV cache", (1, 16, 8, 60, 104), 4)):
    lat = torch.randn(shape, device=dev, generator=g)
    res = {}
    for flag in ("0", "1", "0", "1"):
        os.environ["LCV_DENOISE_GRAPH"] = flag
        torch.cuda.synchronize(); t0 = time.perf_counter()
        with torch.no_grad():
            out = pipe.denoise(lat, pe, pm, ne, nm, num_cond_latents=ncond, num_inference_steps=20, guidance_scale=4.0, use_kv_cache=True)
        torch.cuda.synchronize()
        res.setdefault(flag, []).append(((time.perf_counter() - t0) / 20, out))
    e, gr = min(t for t, _ in res["0"]), min(t for t, _ in res["1"])
    print(f"{name}: eager {e * 1e3:.1f} ms / step, graph replay (capture included, 20 steps) {gr * 1e3:.1f} ms / step ({e / gr:.2f}x); "
          f"bit-equal {torch.equal(res['0'][0][1], res['1'][0][1])}", flush=True)
