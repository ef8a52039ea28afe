#!/usr/bin/env python3
"""The reference's published operating point (480p, 4 clean + 4 noised latent frames, KV cache, CFG): N denoise steps for
profiling (`rocprofv3 --kernel-trace --stats -- python3 tools/ref_point_denoise.py 6`)."""
import sys
import time
from pathlib import Path
ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT / "longcat-video-tta_amd"))
import torch
from longcat_video.modules.longcat_video_dit import LongCatVideoTransformer3DModel
from longcat_video.modules.scheduling_flow_match_euler_discrete import FlowMatchEulerDiscreteScheduler
from longcat_video.pipeline_longcat_video import LongCatVideoPipeline
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 6
dev = torch.device("cuda", 0)
dit = LongCatVideoTransformer3DModel(device=dev, dtype=torch.bfloat16).eval().init_synthetic_(1234)
pipe = LongCatVideoPipeline(scheduler=FlowMatchEulerDiscreteScheduler(), dit=dit); pipe.device = dev
g = torch.Generator(device=dev).manual_seed(1)
pe = torch.randn((1, 1, 512, 4096), generator=g, device=dev).to(torch.bfloat16); ne = torch.randn_like(pe)
pm = torch.zeros((1, 512), dtype=torch.int64, device=dev); pm[:, :77] = 1
x = torch.randn((1, 16, 8, 60, 104), generator=g, device=dev)
kw = dict(num_cond_latents=4, num_inference_steps=50, guidance_scale=4.0, use_kv_cache=True)
x1 = pipe.denoise(x, pe, pm, ne, pm, start_step=0, stop_step=1, **kw)
torch.cuda.synchronize(); t0 = time.perf_counter()
x1 = pipe.denoise(x, pe, pm, ne, pm, start_step=1, stop_step=1 + steps, **kw)
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / steps
print(f"{dt*1e3:.1f} ms per KV-cached CFG step (6 240 noise tokens x 2, 12 480 keys)")
