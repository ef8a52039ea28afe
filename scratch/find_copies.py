"""Where do the torch copy / fill / add kernels of a LoRA-TTA step come from?  torch.profiler with stacks, depth 2, 720p."""
import sys, functools
from pathlib import Path
ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT / "longcat-video-tta_amd")); sys.path.insert(0, str(ROOT))
import torch
from torch.profiler import profile, ProfilerActivity
from longcat_video.modules.longcat_video_dit import LongCatVideoTransformer3DModel
from tta.lora import inject_lora_into_dit
from tta.inner_loop import finetune_lora_on_conditioning
dev = "cuda"; bf = torch.bfloat16
h, w = 90, 160
dit = LongCatVideoTransformer3DModel(device=dev, dtype=bf, depth=2).eval(); dit.init_synthetic_()
for p in dit.parameters(): p.requires_grad = False
mods = inject_lora_into_dit(dit, rank=8, alpha=16.0, target_modules=["qkv", "proj"])
g = torch.Generator(device=dev).manual_seed(1)
cond = torch.randn(1, 16, 4, h, w, device=dev, generator=g).to(bf); train = torch.randn(1, 16, 3, h, w, device=dev, generator=g).to(bf)
pe = torch.randn(1, 1, 512, 4096, device=dev, generator=g).to(bf); pm = torch.zeros(1, 512, dtype=torch.int64, device=dev); pm[:, :77] = 1
kw = dict(lr=2e-4, warmup_steps=3, device=dev, dtype=bf)
finetune_lora_on_conditioning(dit, mods, cond, train, pe, pm, num_steps=1, **kw)
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True, record_shapes=True) as prof:
    finetune_lora_on_conditioning(dit, mods, cond, train, pe, pm, num_steps=1, **kw)
    torch.cuda.synchronize()
from collections import defaultdict
agg = defaultdict(lambda: [0, 0.0])
for ev in prof.events():
    if ev.name in ("aten::copy_", "aten::fill_", "aten::zero_", "aten::add", "aten::add_", "aten::contiguous", "aten::clone", "aten::_to_copy", "aten::zeros", "aten::mul", "aten::cat"):
        dt = ev.device_time_total if hasattr(ev, "device_time_total") else ev.cuda_time_total
        if dt <= 0:
            continue
        frames = [f for f in (ev.stack or []) if "longcat-video-tta_amd" in f or "/tta/" in f]
        key = (ev.name, str(ev.input_shapes)[:60], " <- ".join(fr.split("longcat-video-tta_amd/")[-1] for fr in frames[:3]))
        agg[key][0] += 1; agg[key][1] += dt
rows = sorted(agg.items(), key=lambda kv: -kv[1][1])
for (name, shp, st), (n, t) in rows[:40]:
    print(f"{t/1e3:8.2f} ms  x{n:4d}  {name:18s} {shp:60s} {st}")
