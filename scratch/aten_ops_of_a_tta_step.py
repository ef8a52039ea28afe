"""Which torch (aten) ops of a LoRA-TTA step launch GPU kernels, by self device time and shape?  torch.profiler, depth 2, 720p."""
import sys
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
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True, with_stack=True) as prof:
    finetune_lora_on_conditioning(dit, mods, cond, train, pe, pm, num_steps=1, **kw)
    torch.cuda.synchronize()
rows = []
for ev in prof.key_averages(group_by_input_shape=True, group_by_stack_n=6):
    t = getattr(ev, "self_device_time_total", 0) or getattr(ev, "self_cuda_time_total", 0)
    if t > 0 and ev.key.startswith("aten::"):
        st = [f for f in (ev.stack or []) if "longcat-video-tta_amd" in f or "/tta/" in f]
        rows.append((t, ev.count, ev.key, str(ev.input_shapes)[:70], " <- ".join(s.split("longcat-video-tta_amd/")[-1][:60] for s in st[:2])))
rows.sort(reverse=True)
tot = sum(r[0] for r in rows)
print(f"aten ops with GPU time: {tot/1e3:.2f} ms in total (2 blocks, 1 step)")
for t, n, k, shp, st in rows[:30]:
    print(f"{t/1e3:8.3f} ms x{n:4d} {k:22s} {shp:70s} {st}")
