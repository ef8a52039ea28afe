#!/usr/bin/env python3
"""N LoRA-TTA inner-loop steps at the K3-TTA shape (720p Tc=4 + Tt=3 latent frames = 25 200 tokens, 48 blocks, r=8 on qkv+proj)
for profiling: `NO_CKPT=1 rocprofv3 --kernel-trace --stats -- python3 tools/tta_steps.py 48 720p`."""
import sys, time, functools
from pathlib import Path
ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT / "longcat-video-tta_amd")); sys.path.insert(0, str(ROOT))
import torch
from torch.utils.checkpoint import checkpoint
from longcat_video.modules.longcat_video_dit import LongCatVideoTransformer3DModel
from tta.lora import inject_lora_into_dit, get_lora_parameters, count_lora_parameters
from tta.inner_loop import finetune_lora_on_conditioning
dev="cuda"; bf=torch.bfloat16
depth=int(sys.argv[1]) if len(sys.argv)>1 else 48
h,w = (90,160) if len(sys.argv)<3 or sys.argv[2]=="720p" else (60,104)
dit=LongCatVideoTransformer3DModel(device=dev,dtype=bf,depth=depth).eval(); dit.init_synthetic_()
for p in dit.parameters(): p.requires_grad=False
import os
if os.environ.get('CKPT')=='1':   # block checkpointing only on request (the inner loop's own choice at this size is off)
    dit.gradient_checkpointing=True; dit._gradient_checkpointing_func=functools.partial(checkpoint,use_reentrant=False)
mods=inject_lora_into_dit(dit,rank=8,alpha=16.0,target_modules=["qkv","proj"])
print(count_lora_parameters(mods))
g=torch.Generator(device=dev).manual_seed(1)
cond=torch.randn(1,16,4,h,w,device=dev,generator=g).to(bf); train=torch.randn(1,16,3,h,w,device=dev,generator=g).to(bf)
pe=torch.randn(1,1,512,4096,device=dev,generator=g).to(bf); pm=torch.zeros(1,512,dtype=torch.int64,device=dev); pm[:,:77]=1
torch.cuda.synchronize(); t0=time.time()
res=finetune_lora_on_conditioning(dit,mods,cond,train,pe,pm,num_steps=4,lr=2e-4,warmup_steps=3,device=dev,dtype=bf)
torch.cuda.synchronize()
print("losses",res["losses"],"train_time",res["train_time"],"s/step",res["train_time"]/4, "peak GB", torch.cuda.max_memory_allocated()/2**30)
