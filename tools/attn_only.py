#!/usr/bin/env python3
"""One K3-shaped self-attention launch (B=2 CFG batch, H=32, N=46 800, D=128) for PMC collection under rocprofv3."""
import sys
from pathlib import Path
ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT / "longcat-video-tta_amd"))
import torch
from lcv_hip import ops
B, N, H, D = 2, 46800, 32, 128
g = torch.Generator(device="cuda").manual_seed(0)
qkv = torch.randn(B, N, 3, H, D, device="cuda", generator=g, dtype=torch.float32).to(torch.bfloat16)
o = torch.empty(B, N, H, D, device="cuda", dtype=torch.bfloat16)
# the production call pattern of the self-attention path: q pre-scaled into log2 units, scale = ln 2 (attn_fwd_kernel<8,0,false,3>)
q = (qkv[:, :, 0].float() * ops.log2_qscale(D ** -0.5)).to(torch.bfloat16)
for _ in range(2):
    ops.attention(q, qkv[:, :, 1], qkv[:, :, 2], ops.LN2, out=o)
torch.cuda.synchronize()
print("done", float(o.float().abs().mean()))
