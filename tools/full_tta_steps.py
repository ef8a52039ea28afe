#!/usr/bin/env python3
"""N full-model TTA inner steps (every DiT parameter trainable, block checkpointing on, fused clip + SGD / AdamW) at full
width and depth: `python tools/full_tta_steps.py [depth=48] [480p|720p] [sgd|adamw]`.  480p: Tc=3 + Tt=1 latent frames
(6 240 tokens, the reference's operating point); 720p: Tc=4 + Tt=3 (25 200 tokens)."""
import functools
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT / "longcat-video-tta_amd")); sys.path.insert(0, str(ROOT))
import torch  # noqa: E402
from torch.utils.checkpoint import checkpoint  # noqa: E402

from longcat_video.modules.longcat_video_dit import LongCatVideoTransformer3DModel  # noqa: E402
from tta.full_tta import finetune_full_on_conditioning  # noqa: E402

dev, bf = "cuda", torch.bfloat16
depth = int(sys.argv[1]) if len(sys.argv) > 1 else 48
res = sys.argv[2] if len(sys.argv) > 2 else "480p"
opt = sys.argv[3] if len(sys.argv) > 3 else "sgd"
(h, w), (tc, tt) = ((90, 160), (4, 3)) if res == "720p" else ((60, 104), (3, 1))
dit = LongCatVideoTransformer3DModel(device=dev, dtype=bf, depth=depth).eval().init_synthetic_()
dit.gradient_checkpointing = True
dit._gradient_checkpointing_func = functools.partial(checkpoint, use_reentrant=False)
for p in dit.parameters():
    p.requires_grad = True
print("trainable params:", sum(p.numel() for p in dit.parameters()), flush=True)
g = torch.Generator(device=dev).manual_seed(1)
cond = torch.randn(1, 16, tc, h, w, device=dev, generator=g).to(bf)
train = torch.randn(1, 16, tt, h, w, device=dev, generator=g).to(bf)
pe = torch.randn(1, 1, 512, 4096, device=dev, generator=g).to(bf)
pm = torch.zeros(1, 512, dtype=torch.int64, device=dev); pm[:, :77] = 1
n = 3
r = finetune_full_on_conditioning(dit, cond, train, pe, pm, num_steps=1, lr=1e-5, warmup_steps=0, device=dev, dtype=bf, optimizer_type=opt)
torch.cuda.synchronize()
r = finetune_full_on_conditioning(dit, cond, train, pe, pm, num_steps=n, lr=1e-5, warmup_steps=2, device=dev, dtype=bf, optimizer_type=opt)
print(f"{res} {opt}: losses {r['losses']} | {r['train_time'] / n:.2f} s/step | peak {torch.cuda.max_memory_allocated() / 2**30:.1f} GiB", flush=True)
